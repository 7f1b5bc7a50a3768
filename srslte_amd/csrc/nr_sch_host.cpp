// nr_sch_host.cpp -- C ABI of NR LDPC rate matching and the LDPC encoder (include/srsran_amd/phy_nr_sch_abi.h).
#include "hip_common.h"
#include "nr_sch_device.h"
#include "srsran_amd/phy_nr_sch_abi.h"
#include "tables/nr_ldpc_bg_table.h"

#include <map>
#include <vector>

using namespace phyhip;

extern "C" const uint8_t LSindex[385]; // ldpc_host.cpp (base_graph.c:50)

namespace {

inline uint32_t mod_bits(int m) // srsran_mod_bits_x_symbol, phy_common.c:264-280
{
  static const uint32_t q[5] = {1, 2, 4, 6, 8};
  return (m >= 0 && m <= 4) ? q[m] : 0;
}

// init_rm, ldpc_rm.c:113-167.  Returns false (after the reference's message) where the reference fails.
struct RmCfg {
  uint32_t N, K, Ncb, k0, Qm;
};
bool rm_cfg(uint32_t E, int bg, uint32_t ls, uint32_t rv, int mod, uint32_t Nref, RmCfg* c)
{
  static const uint32_t basek0[4][2] = {{0, 0}, {17, 13}, {33, 25}, {56, 43}}; // TS 38.212 table 5.4.2.1-2, in units of Z
  const uint32_t        maxE         = 273 * 13 * 12 * 8 * 4;                    // ldpc_rm.c:77
  if (bg < 0 || bg > 1 || rv > 3) {
    return false;
  }
  c->Qm = mod_bits(mod);
  c->N  = ls * (bg == 0 ? 66u : 50u);
  c->K  = ls * (bg == 0 ? 22u : 10u);
  if (E > maxE) {
    fprintf(stderr, "Wrong RM codeword length (E) = %d. It must be smaller than %d for base graph %d\n", E, maxE, bg + 1);
    return false;
  }
  if (c->Qm == 0) {
    fprintf(stderr, "Invalid modulation order\n");
    return false;
  }
  if (E % c->Qm) {
    fprintf(stderr, "Wrong RM codeword length (E) = %d. It must be a multiple of modulation order = %d\n", E, c->Qm);
    return false;
  }
  if (c->N <= Nref) {
    c->Ncb = c->N;
    c->k0  = ls * basek0[rv][bg];
  } else {
    c->Ncb = Nref;
    c->k0  = ls * ((basek0[rv][bg] * Nref) / c->N);
  }
  return true;
}

// ---- device copy of one (base graph, lifting size) + the encoder's solving order ----------------------------------------------
struct Graph {
  int              bgN = 0, bgM = 0, bgK = 0, Z = 0;
  int*             d_row_start = nullptr;
  int*             d_edges     = nullptr;
  int              a           = 0;
  nrsch::EncStep   step[3];
};

bool build_graph(int bg, int Z, Graph* g)
{
  const int ils = Z >= 2 && Z <= 384 ? LSindex[Z] : 255;
  if (bg < 0 || bg > 1 || ils > 7) {
    fprintf(stderr, "Invalid lifting size %d\n", Z);
    return false;
  }
  const nr_ldpc_edge_t* edges = bg == 0 ? nr_ldpc_bg1_edges : nr_ldpc_bg2_edges;
  const int             E     = bg == 0 ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
  g->bgN = bg == 0 ? 68 : 52;
  g->bgM = bg == 0 ? 46 : 42;
  g->bgK = g->bgN - g->bgM;
  g->Z   = Z;
  std::vector<int> rs(g->bgM + 1, 0), ed(E);
  for (int e = 0, row = 0; e < E; e++) {
    while (row < edges[e].row) {
      rs[++row] = e;
    }
    ed[e] = edges[e].col | ((edges[e].v[ils] % Z) << 8);
  }
  for (int row = edges[E - 1].row; row < g->bgM;) {
    rs[++row] = E;
  }
  // core (rows 0..3 x parity columns bgK..bgK+3): column bgK has three entries, two of them with equal shifts, so the sum
  // of the four rows leaves one rotation of the first parity block; the other blocks follow row by row
  std::vector<int> sh0;
  for (int e = rs[0]; e < rs[4]; e++) {
    if ((ed[e] & 0xff) == g->bgK) {
      sh0.push_back(ed[e] >> 8);
    }
  }
  g->a = -1;
  for (int s : sh0) {
    int c = 0;
    for (int t : sh0) {
      c += t == s;
    }
    if (c & 1) {
      g->a = s;
    }
  }
  if (g->a < 0) {
    return false;
  }
  bool known[4] = {true, false, false, false};
  int  n_steps  = 0;
  for (int pass = 0; pass < 4 && n_steps < 3; pass++) {
    for (int m = 0; m < 4 && n_steps < 3; m++) {
      int            unk = -1, ush = 0, nunk = 0;
      nrsch::EncStep st{};
      for (int e = rs[m]; e < rs[m + 1]; e++) {
        const int c = (ed[e] & 0xff) - g->bgK;
        if (c < 0 || c > 3) {
          continue;
        }
        if (!known[c]) {
          unk = c;
          ush = ed[e] >> 8;
          nunk++;
        } else if (st.n_terms < NRSCH_MAX_CORE_TERMS) {
          st.blk[st.n_terms]  = c;
          st.sh[st.n_terms++] = ed[e] >> 8;
        } else {
          return false;
        }
      }
      if (nunk != 1) {
        continue;
      }
      st.row = m;
      st.unk = unk;
      st.ush = ush;
      g->step[n_steps++] = st;
      known[unk]         = true;
    }
  }
  if (n_steps != 3) {
    return false;
  }
  if (hipMalloc(&g->d_row_start, rs.size() * sizeof(int)) != hipSuccess || hipMalloc(&g->d_edges, ed.size() * sizeof(int)) != hipSuccess ||
      upload(g->d_row_start, rs.data(), rs.size() * sizeof(int)) != hipSuccess ||
      upload(g->d_edges, ed.data(), ed.size() * sizeof(int)) != hipSuccess) {
    set_error("nr_sch: cannot put the base graph on the device");
    return false;
  }
  return true;
}

// clamping / rounding of cdwd_rm_length -> number of check rows to encode (ldpc_encoder.c:63-78,92)
uint32_t enc_layers(const Graph& g, uint32_t cdwd_rm_length)
{
  const uint32_t Z = g.Z, full = (uint32_t)(g.bgN - 2) * Z;
  cdwd_rm_length = std::min(cdwd_rm_length, full);
  cdwd_rm_length = std::max(cdwd_rm_length, (uint32_t)(g.bgK + 2) * Z);
  if (cdwd_rm_length % Z) {
    cdwd_rm_length = (cdwd_rm_length / Z + 1) * Z;
  }
  return cdwd_rm_length / Z - g.bgK + 2;
}

} // namespace

// ------------------------------------------------------------------------------------------------ batch object
struct srsran_hip_nr_sch {
  DeviceTag tag;
  std::map<uint32_t, Graph> graphs; // bg << 16 | Z
  nrsch::CbJob*             d_jobs  = nullptr;
  nrsch::CbJob*             h_jobs  = nullptr; // pinned
  size_t                    cap     = 0;
  hipEvent_t                done    = nullptr;
  bool                      pending = false;

  const Graph* graph(int bg, int Z)
  {
    const uint32_t key = ((uint32_t)bg << 16) | (uint32_t)Z;
    auto           it  = graphs.find(key);
    if (it == graphs.end()) {
      Graph g;
      if (!build_graph(bg, Z, &g)) {
        return nullptr;
      }
      it = graphs.emplace(key, g).first;
    }
    return &it->second;
  }

  // job list -> device (the previous call's kernel must have consumed the buffer first)
  int stage(const srsran_hip_ldpc_cb_t* cbs, uint32_t n, const Graph* enc, hipStream_t st, const uint8_t* new_data = nullptr)
  {
    if (pending) {
      PHY_HIP_CHECK(hipEventSynchronize(done), SRSRAN_ERROR);
      pending = false;
    }
    if (n > cap) {
      (void)hipFree(d_jobs);
      (void)hipHostFree(h_jobs);
      d_jobs = nullptr;
      h_jobs = nullptr;
      cap    = 0;
      const size_t c = (size_t)n + n / 2 + 16;
      PHY_HIP_CHECK(hipMalloc(&d_jobs, c * sizeof(nrsch::CbJob)), SRSRAN_ERROR);
      PHY_HIP_CHECK(hipHostMalloc(&h_jobs, c * sizeof(nrsch::CbJob)), SRSRAN_ERROR);
      cap = c;
    }
    for (uint32_t i = 0; i < n; i++) {
      h_jobs[i] = nrsch::CbJob{cbs[i].in_offset, cbs[i].out_offset, cbs[i].E, enc ? enc_layers(*enc, cbs[i].E) : ((new_data && new_data[i]) ? 1u : 0u)};
    }
    PHY_HIP_CHECK(hipMemcpyAsync(d_jobs, h_jobs, n * sizeof(nrsch::CbJob), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    return SRSRAN_SUCCESS;
  }
  int finish(hipStream_t st)
  {
    PHY_HIP_CHECK(hipEventRecord(done, st), SRSRAN_ERROR);
    pending = true;
    return SRSRAN_SUCCESS;
  }
};

extern "C" int srsran_hip_nr_sch_create(srsran_hip_nr_sch_t** hh)
{
  if (!hh) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  srsran_hip_nr_sch* h = new srsran_hip_nr_sch;
  if (hipEventCreateWithFlags(&h->done, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return SRSRAN_ERROR;
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_nr_sch_free(srsran_hip_nr_sch_t* h)
{
  if (!h) {
    return;
  }
  if (h->pending) {
    (void)hipEventSynchronize(h->done);
  }
  for (auto& kv : h->graphs) {
    (void)hipFree(kv.second.d_row_start);
    (void)hipFree(kv.second.d_edges);
  }
  (void)hipFree(h->d_jobs);
  (void)hipHostFree(h->h_jobs);
  (void)hipEventDestroy(h->done);
  delete h;
}

static int rm_batch(srsran_hip_nr_sch_t* h, bool tx, int llr_type, const void* d_in, void* d_out, const srsran_hip_ldpc_cb_t* cbs, uint32_t n_cb,
                    uint32_t F, int bg, uint32_t ls, uint32_t rv, int mod, uint32_t Nref, hipStream_t st, const uint8_t* new_data = nullptr)
{
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_ldpc_rm_batch", SRSRAN_ERROR);
  }
  if (!h || (n_cb && (!cbs || !d_in || !d_out)) || llr_type < 0 || llr_type > 2) {
    set_error("ldpc rate matching: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (n_cb == 0) {
    return SRSRAN_SUCCESS;
  }
  if (n_cb > 65535) { // one launch addresses its code blocks through grid.y: larger batches go in pieces, in stream order
    for (uint32_t i = 0; i < n_cb; i += 65535) {
      const int rc = rm_batch(h, tx, llr_type, d_in, d_out, cbs + i, std::min(65535u, n_cb - i), F, bg, ls, rv, mod, Nref, st, new_data ? new_data + i : nullptr);
      if (rc != SRSRAN_SUCCESS) {
        return rc;
      }
    }
    return SRSRAN_SUCCESS;
  }
  RmCfg    c;
  uint32_t max_E = 0;
  for (uint32_t i = 0; i < n_cb; i++) {
    if (!rm_cfg(cbs[i].E, bg, ls, rv, mod, Nref, &c)) {
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    max_E = std::max(max_E, cbs[i].E);
  }
  if (F > c.K - 2 * ls || c.Ncb == 0 || c.Ncb > 65535) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (h->stage(cbs, n_cb, nullptr, st, tx ? nullptr : new_data) != SRSRAN_SUCCESS) {
    return SRSRAN_ERROR;
  }
  uint32_t min_E_new = ~0u;
  for (uint32_t i = 0; !tx && new_data && i < n_cb; i++) {
    if (new_data[i]) {
      min_E_new = std::min(min_E_new, cbs[i].E);
    }
  }
  nrsch::RmParams p{};
  p.in     = d_in;
  p.out    = d_out;
  p.jobs   = h->d_jobs;
  p.n_cb   = n_cb;
  p.Ncb    = c.Ncb;
  p.k0     = c.k0;
  p.end_ex = c.K - 2 * ls;
  p.ini_ex = p.end_ex - F;
  p.Qm     = c.Qm;
  p.type   = llr_type == SRSRAN_HIP_LLR_BYTE ? nrsch::T_I8 : (llr_type == SRSRAN_HIP_LLR_SHORT ? nrsch::T_I16 : nrsch::T_F32);
  PHY_HIP_CHECK(tx ? nrsch::launch_rm_tx(p, st) : nrsch::launch_rm_rx(p, max_E, st, min_E_new), SRSRAN_ERROR);
  return h->finish(st);
}

extern "C" int srsran_hip_ldpc_rm_rx_batch(srsran_hip_nr_sch_t* h, int llr_type, const void* d_in, void* d_softbuf, const srsran_hip_ldpc_cb_t* cbs,
                                           uint32_t n_cb, uint32_t F, srsran_basegraph_t bg, uint32_t ls, uint32_t rv, srsran_mod_t mod_type,
                                           uint32_t Nref, void* stream)
{
  return rm_batch(h, false, llr_type, d_in, d_softbuf, cbs, n_cb, F, (int)bg, ls, rv, (int)mod_type, Nref, (hipStream_t)stream);
}

// the same with a flag per code block: 1 = new data, its soft-buffer row is overwritten (what srsran_softbuffer_rx_reset + the first
// srsran_ldpc_rm_rx_c leave there) instead of accumulated into
extern "C" int srsran_hip_ldpc_rm_rx_batch_new(srsran_hip_nr_sch_t* h, int llr_type, const void* d_in, void* d_softbuf, const srsran_hip_ldpc_cb_t* cbs,
                                               const uint8_t* new_data, uint32_t n_cb, uint32_t F, srsran_basegraph_t bg, uint32_t ls, uint32_t rv,
                                               srsran_mod_t mod_type, uint32_t Nref, void* stream)
{
  return rm_batch(h, false, llr_type, d_in, d_softbuf, cbs, n_cb, F, (int)bg, ls, rv, (int)mod_type, Nref, (hipStream_t)stream, new_data);
}

extern "C" int srsran_hip_ldpc_rm_tx_batch(srsran_hip_nr_sch_t* h, const uint8_t* d_codewords, uint8_t* d_out, const srsran_hip_ldpc_cb_t* cbs,
                                           uint32_t n_cb, srsran_basegraph_t bg, uint32_t ls, uint32_t rv, srsran_mod_t mod_type, uint32_t Nref,
                                           void* stream)
{
  return rm_batch(h, true, SRSRAN_HIP_LLR_BYTE, d_codewords, d_out, cbs, n_cb, 0, (int)bg, ls, rv, (int)mod_type, Nref, (hipStream_t)stream);
}

extern "C" int srsran_hip_ldpc_encode_batch(srsran_hip_nr_sch_t* h, const uint8_t* d_messages, uint8_t* d_codewords, const srsran_hip_ldpc_cb_t* cbs,
                                            uint32_t n_cb, srsran_basegraph_t bg, uint32_t ls, void* stream)
{
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_ldpc_encode_batch", SRSRAN_ERROR);
  }
  if (!h || (n_cb && (!cbs || !d_messages || !d_codewords))) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (n_cb == 0) {
    return SRSRAN_SUCCESS;
  }
  const Graph* g = h->graph((int)bg, (int)ls);
  if (!g) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t st = (hipStream_t)stream;
  if (h->stage(cbs, n_cb, g, st) != SRSRAN_SUCCESS) {
    return SRSRAN_ERROR;
  }
  nrsch::EncParams p{};
  p.in        = d_messages;
  p.out       = d_codewords;
  p.jobs      = h->d_jobs;
  p.n_cb      = n_cb;
  p.row_start = g->d_row_start;
  p.edges     = g->d_edges;
  p.Z         = g->Z;
  p.bgN       = g->bgN;
  p.bgM       = g->bgM;
  p.bgK       = g->bgK;
  p.a         = g->a;
  for (int i = 0; i < 3; i++) {
    p.step[i] = g->step[i];
  }
  PHY_HIP_CHECK(nrsch::launch_encode(p, st), SRSRAN_ERROR);
  return h->finish(st);
}

// ------------------------------------------------------------------------------------------------ drop-in objects (host pointers)
namespace {

struct HostCtx { // behind srsran_ldpc_rm_t.ptr / srsran_ldpc_encoder_t.ptr
  DeviceTag tag;
  srsran_hip_nr_sch_t* h     = nullptr;
  hipStream_t          st    = nullptr;
  void*                d_in  = nullptr;
  void*                d_out = nullptr;
  size_t               cap_in = 0, cap_out = 0;
};

HostCtx* ctx_new()
{
  HostCtx* c = new HostCtx;
  if (srsran_hip_nr_sch_create(&c->h) != SRSRAN_SUCCESS || hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) {
    srsran_hip_nr_sch_free(c->h);
    delete c;
    return nullptr;
  }
  return c;
}

void ctx_free(HostCtx* c)
{
  if (!c) {
    return;
  }
  srsran_hip_nr_sch_free(c->h);
  (void)hipFree(c->d_in);
  (void)hipFree(c->d_out);
  (void)hipStreamDestroy(c->st);
  delete c;
}

bool ctx_grow(void** p, size_t* cap, size_t need)
{
  if (need <= *cap) {
    return true;
  }
  (void)hipFree(*p);
  *p   = nullptr;
  *cap = 0;
  if (hipMalloc(p, need + 256) != hipSuccess) {
    return false;
  }
  *cap = need + 256;
  return true;
}

int rm_init(srsran_ldpc_rm_t* q)
{
  if (!q) {
    return -1;
  }
  memset(q, 0, sizeof(*q));
  q->ptr = ctx_new();
  if (!q->ptr) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_rm init: %s (there is no CPU fallback)\n", get_error());
    return -1;
  }
  return 0;
}

void rm_free(srsran_ldpc_rm_t* q)
{
  if (q) {
    ctx_free((HostCtx*)q->ptr);
    q->ptr = nullptr;
  }
}

int rm_host(srsran_ldpc_rm_t* q, bool tx, int llr_type, size_t es, const void* input, void* output, uint32_t E, uint32_t F, int bg, uint32_t ls,
            uint32_t rv, int mod, uint32_t Nref)
{
  RmCfg c;
  if (!q || !q->ptr || !rm_cfg(E, bg, ls, rv, mod, Nref, &c)) {
    return -1;
  }
  q->N = c.N, q->E = E, q->K = c.K, q->F = F, q->ls = (uint16_t)ls, q->mod_order = c.Qm, q->bg = (srsran_basegraph_t)bg, q->Ncb = c.Ncb, q->k0 = c.k0;
  HostCtx*     x        = (HostCtx*)q->ptr;
  PHY_DEV_GUARD(x->tag, "srsran_ldpc_rm", -1);
  const size_t in_bytes = tx ? c.N : E * es, out_bytes = tx ? E : c.N * es;
  if (!ctx_grow(&x->d_in, &x->cap_in, in_bytes) || !ctx_grow(&x->d_out, &x->cap_out, out_bytes)) {
    return -1;
  }
  srsran_hip_ldpc_cb_t cb = {0, 0, E};
  PHY_HIP_CHECK(hipMemcpyAsync(x->d_in, input, in_bytes, hipMemcpyHostToDevice, x->st), -1);
  if (!tx) {
    PHY_HIP_CHECK(hipMemcpyAsync(x->d_out, output, out_bytes, hipMemcpyHostToDevice, x->st), -1);
  }
  if (rm_batch(x->h, tx, llr_type, x->d_in, x->d_out, &cb, 1, F, bg, ls, rv, mod, Nref, x->st) != SRSRAN_SUCCESS) {
    return -1;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(output, x->d_out, out_bytes, hipMemcpyDeviceToHost, x->st), -1);
  PHY_HIP_CHECK(hipStreamSynchronize(x->st), -1);
  return 0;
}

} // namespace

extern "C" int srsran_ldpc_rm_tx_init(srsran_ldpc_rm_t* q)
{
  return rm_init(q);
}
extern "C" int srsran_ldpc_rm_rx_init_f(srsran_ldpc_rm_t* q)
{
  return rm_init(q);
}
extern "C" int srsran_ldpc_rm_rx_init_s(srsran_ldpc_rm_t* q)
{
  return rm_init(q);
}
extern "C" int srsran_ldpc_rm_rx_init_c(srsran_ldpc_rm_t* q)
{
  return rm_init(q);
}
extern "C" void srsran_ldpc_rm_tx_free(srsran_ldpc_rm_t* q)
{
  rm_free(q);
}
extern "C" void srsran_ldpc_rm_rx_free_f(srsran_ldpc_rm_t* q)
{
  rm_free(q);
}
extern "C" void srsran_ldpc_rm_rx_free_s(srsran_ldpc_rm_t* q)
{
  rm_free(q);
}
extern "C" void srsran_ldpc_rm_rx_free_c(srsran_ldpc_rm_t* q)
{
  rm_free(q);
}

extern "C" int srsran_ldpc_rm_tx(srsran_ldpc_rm_t* q, const uint8_t* input, uint8_t* output, const uint32_t E, const srsran_basegraph_t bg,
                                 const uint32_t ls, const uint8_t rv, const srsran_mod_t mod_type, const uint32_t Nref)
{
  return rm_host(q, true, SRSRAN_HIP_LLR_BYTE, 1, input, output, E, 0, (int)bg, ls, rv, (int)mod_type, Nref);
}
extern "C" int srsran_ldpc_rm_rx_f(srsran_ldpc_rm_t* q, const float* input, float* output, const uint32_t E, const uint32_t F,
                                   const srsran_basegraph_t bg, const uint32_t ls, const uint8_t rv, const srsran_mod_t mod_type, const uint32_t Nref)
{
  return rm_host(q, false, SRSRAN_HIP_LLR_FLOAT, 4, input, output, E, F, (int)bg, ls, rv, (int)mod_type, Nref);
}
extern "C" int srsran_ldpc_rm_rx_s(srsran_ldpc_rm_t* q, const int16_t* input, int16_t* output, const uint32_t E, const uint32_t F,
                                   const srsran_basegraph_t bg, const uint32_t ls, const uint8_t rv, const srsran_mod_t mod_type, const uint32_t Nref)
{
  return rm_host(q, false, SRSRAN_HIP_LLR_SHORT, 2, input, output, E, F, (int)bg, ls, rv, (int)mod_type, Nref);
}
extern "C" int srsran_ldpc_rm_rx_c(srsran_ldpc_rm_t* q, const int8_t* input, int8_t* output, const uint32_t E, const uint32_t F,
                                   const srsran_basegraph_t bg, const uint32_t ls, const uint8_t rv, const srsran_mod_t mod_type, const uint32_t Nref)
{
  if (rm_host(q, false, SRSRAN_HIP_LLR_BYTE, 1, input, output, E, F, (int)bg, ls, rv, (int)mod_type, Nref) != 0) {
    return -1;
  }
  return (int)std::min(q->k0 + q->E, q->Ncb); // ldpc_rm.c:704-705
}

// ---- encoder object
static int enc_encode(void* o, const uint8_t* input, uint8_t* output, uint32_t input_length, uint32_t cdwd_rm_length)
{
  srsran_ldpc_encoder_t* q = (srsran_ldpc_encoder_t*)o;
  HostCtx*               x = (HostCtx*)q->ptr;
  if (x) {
    PHY_DEV_GUARD(x->tag, "srsran_ldpc_encoder_encode", -1);
  }
  if (input_length / q->bgK != q->ls) {
    fprintf(stderr, "Dimension mismatch.\n"); // ldpc_encoder.c:59-62
    return -1;
  }
  const Graph* g = x->h->graph((int)q->bg, q->ls);
  if (!g) {
    return -1;
  }
  const size_t in_bytes = (size_t)q->liftK;
  const size_t written  = (size_t)(enc_layers(*g, cdwd_rm_length) + q->bgK - 2) * q->ls; // everything beyond stays as it was
  if (!ctx_grow(&x->d_in, &x->cap_in, in_bytes) || !ctx_grow(&x->d_out, &x->cap_out, (size_t)(q->liftN - 2 * q->ls))) {
    return -1;
  }
  srsran_hip_ldpc_cb_t cb = {0, 0, cdwd_rm_length};
  PHY_HIP_CHECK(hipMemcpyAsync(x->d_in, input, in_bytes, hipMemcpyHostToDevice, x->st), -1);
  if (srsran_hip_ldpc_encode_batch(x->h, (const uint8_t*)x->d_in, (uint8_t*)x->d_out, &cb, 1, q->bg, q->ls, x->st) != SRSRAN_SUCCESS) {
    return -1;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(output, x->d_out, written, hipMemcpyDeviceToHost, x->st), -1);
  PHY_HIP_CHECK(hipStreamSynchronize(x->st), -1);
  return 0;
}

static void enc_free(void* o)
{
  srsran_ldpc_encoder_t* q = (srsran_ldpc_encoder_t*)o;
  ctx_free((HostCtx*)q->ptr);
  free(q->pcm);
}

extern "C" int srsran_ldpc_encoder_init(srsran_ldpc_encoder_t* q, srsran_ldpc_encoder_type_t type, srsran_basegraph_t bg, uint16_t ls)
{
  if (!q) {
    return -1;
  }
  memset(q, 0, sizeof(*q));
  if (bg != BG1 && bg != BG2) {
    fprintf(stderr, "Base Graph BG%d does not exist\n", (int)bg + 1); // ldpc_encoder.c:525-527
    return -1;
  }
  if ((int)type < 0 || (int)type > SRSRAN_LDPC_ENCODER_AVX512) {
    return -1;
  }
  q->bgN   = bg == BG1 ? 68 : 52;
  q->bgM   = bg == BG1 ? 46 : 42;
  q->bg    = bg;
  q->bgK   = q->bgN - q->bgM;
  q->ls    = ls;
  q->liftK = ls * q->bgK;
  q->liftM = ls * q->bgM;
  q->liftN = ls * q->bgN;
  q->pcm   = (uint16_t*)malloc((size_t)q->bgM * q->bgN * sizeof(uint16_t));
  if (!q->pcm || create_compact_pcm(q->pcm, NULL, bg, ls) != 0) { // prints "Invalid lifting size" like the reference
    free(q->pcm);
    memset(q, 0, sizeof(*q));
    return -1;
  }
  HostCtx* x = ctx_new();
  if (!x || !x->h->graph((int)bg, ls)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_encoder_init: %s (there is no CPU fallback)\n", get_error());
    ctx_free(x);
    free(q->pcm);
    memset(q, 0, sizeof(*q));
    return -1;
  }
  q->ptr    = x;
  q->free   = enc_free;
  q->encode = enc_encode;
  return 0;
}

extern "C" void srsran_ldpc_encoder_free(srsran_ldpc_encoder_t* q)
{
  if (!q) {
    return;
  }
  if (q->free) {
    q->free(q);
  }
  memset(q, 0, sizeof(*q));
}

extern "C" int srsran_ldpc_encoder_encode(srsran_ldpc_encoder_t* q, const uint8_t* input, uint8_t* output, uint32_t input_length)
{
  return q->encode(q, input, output, input_length, q->liftN - 2 * q->ls);
}

extern "C" int srsran_ldpc_encoder_encode_rm(srsran_ldpc_encoder_t* q, const uint8_t* input, uint8_t* output, uint32_t input_length,
                                             uint32_t cdwd_rm_length)
{
  return q->encode(q, input, output, input_length, cdwd_rm_length);
}
