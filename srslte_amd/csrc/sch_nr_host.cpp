// sch_nr_host.cpp -- NR shared-channel receive side for batches of transport blocks: the loop of sch_nr_decode
// (lib/src/phy/phch/sch_nr.c:522-713) with every code block of every transport block of a call on the device at once.
//   segmentation            srsran_cbsegm_ldpc_bg1/2 (cbsegm.c:159-285), srsran_sch_nr_fill_tb_info (sch_nr.c:77-144)
//   rate de-matching        srsran_ldpc_rm_rx_c into the soft buffer (HARQ combining), one launch per (graph, Z, rv, Qm, Nref, F)
//   decoding                srsran_ldpc_decoder_decode_crc_c with CRC24B / CRC24A / CRC16 early stop, one launch per (graph, Z, n_llr)
//   code-block verdict      CRC matched and not all zero (:633-639), bits packed into the per-block data rows (:650-652)
//   transport block         payload assembly and transport CRC (:667-705)
// The only value taken from the caller instead of being derived is Nref (:119-126 compute it from the carrier with
// srsran_ra_nr_tbs, the reference's resource-allocation code, which is outside this library).
#include "hip_common.h"
#include "nr_sch_device.h"
#include "srsran_amd/phy_batch.h"
#include "srsran_amd/phy_modem_abi.h"
#include "srsran_amd/phy_nr_sch_abi.h"
#include "srsran_amd/phy_sch_abi.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <tuple>
#include <vector>

using namespace phyhip;

extern "C" const uint8_t LSindex[385]; // ldpc_host.cpp (base_graph.c:50)

namespace {

struct TbCfg { // srsran_sch_nr_tb_info_t
  int      bg;
  uint32_t Qm, A, L_tb, L_cb, B, Bp, Kp, Kr, F, Z, G, Nl, Nref, C, N;
};

// TS 38.212 5.2.2 as srsran_cbsegm_ldpc (cbsegm.c:201-275): transport CRC length, number of code blocks, lifting size
bool nr_segment(int bg, uint32_t tbs, uint32_t* L_tb, uint32_t* C_out, uint32_t* Z_out)
{
  const uint32_t L = tbs <= 3824 ? 16 : 24, K_cb = bg == 0 ? 8448 : 3840, B = tbs + L;
  uint32_t       C, Bp;
  if (B <= K_cb) { // cbsegm_cb_size, cbsegm.c:51-60
    C  = 1;
    Bp = B;
  } else {
    C  = (B + (K_cb - 24) - 1) / (K_cb - 24);
    Bp = B + 24 * C;
  }
  const uint32_t Kp  = Bp / C;
  uint32_t       K_b = 22;
  if (bg == 1) {
    K_b = B > 640 ? 10 : (B > 560 ? 9 : (B > 192 ? 8 : 6));
  }
  uint32_t Z = (Kp + K_b - 1) / K_b;
  while (Z <= 384 && LSindex[Z] == 0xFF) { // cbsegm_ldpc_select_ls (VOID_LIFTSIZE = 255)
    Z++;
  }
  if (Z > 384) {
    return false;
  }
  *L_tb = L, *C_out = C, *Z_out = Z;
  return true;
}

int cbsegm_ldpc(srsran_cbsegm_t* s, int bg, uint32_t tbs)
{
  if (!s) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  memset(s, 0, sizeof(*s));
  if (tbs == 0) {
    return SRSRAN_SUCCESS;
  }
  uint32_t L, C, Z;
  if (!nr_segment(bg, tbs, &L, &C, &Z)) {
    return SRSRAN_ERROR;
  }
  const uint32_t K = Z * (bg == 0 ? 22u : 10u);
  s->tbs = tbs, s->L_tb = L, s->L_cb = C > 1 ? 24 : 0, s->C = C, s->C1 = C, s->K1 = K, s->F = K * C; // F = K C as the reference has it
  s->K1_idx = LSindex[Z], s->Z = Z;
  return SRSRAN_SUCCESS;
}

// srsran_sch_nr_fill_tb_info with cbsegm_ldpc; false where the reference fails
bool tb_cfg(const srsran_hip_nr_tb_t& tb, TbCfg* c)
{
  static const uint32_t qm[5] = {1, 2, 4, 6, 8};
  if (tb.tbs == 0 || tb.mod > 4 || tb.N_L == 0 || (tb.rv & ~SRSRAN_HIP_NR_TB_NEW_DATA) > 3) {
    return false;
  }
  c->bg = ((tb.tbs <= 292) || (tb.tbs <= 3824 && tb.R <= 0.67) || (tb.R <= 0.25)) ? 1 : 0; // sch_nr.c:35-45
  uint32_t L, C, Z;
  if (!nr_segment(c->bg, tb.tbs, &L, &C, &Z)) {
    return false;
  }
  const uint32_t B = tb.tbs + L;
  c->Qm   = qm[tb.mod];
  c->A    = tb.tbs;
  c->L_tb = L;
  c->L_cb = C > 1 ? 24 : 0;
  c->B    = B;
  c->Bp   = B + c->L_cb * C;
  c->Kp   = c->Bp / C;
  c->Kr   = Z * (c->bg == 0 ? 22u : 10u);
  c->F    = c->Kr - c->Kp;
  c->Z    = Z;
  c->G    = tb.nof_bits;
  c->Nl   = tb.N_L;
  c->N    = Z * (c->bg == 0 ? 66u : 50u);
  c->Nref = tb.Nref ? tb.Nref : c->N;
  c->C    = C;
  return true;
}

uint32_t get_E(const TbCfg& c, uint32_t j) // sch_nr_get_E, sch_nr.c:146-157 (all code blocks transmitted)
{
  const uint32_t q = c.Nl * c.Qm;
  if (j <= (c.C - (c.G / q) % c.C - 1)) {
    return q * (c.G / (q * c.C));
  }
  return q * ((c.G + q * c.C - 1) / (q * c.C));
}

struct Job {
  uint32_t tb, cb, E, in_off, n_llr, cb_len, fresh;
  int      bg;
  uint32_t Z, rv, mod, Nref, F, poly, order;
};

template <class T>
bool ensure(T** d, T** h, size_t* cap, size_t n)
{
  if (n <= *cap) {
    return true;
  }
  (void)hipFree(*d);
  (void)hipHostFree(*h);
  *d = nullptr, *h = nullptr, *cap = 0;
  const size_t c = n + n / 2 + 16;
  if (hipMalloc(d, c * sizeof(T)) != hipSuccess || host_image_alloc(h, c * sizeof(T)) != hipSuccess) {
    return false;
  }
  *cap = c;
  return true;
}

} // namespace

// cbsegm.h:59-67 / cbsegm.c:277-285: NR segmentation for the two base graphs (srsran_sch_nr_fill_tb_info and the reference's own NR
// objects call them when cbsegm.c is dropped from the link)
extern "C" int srsran_cbsegm_ldpc_bg1(srsran_cbsegm_t* s, uint32_t tbs)
{
  return cbsegm_ldpc(s, 0, tbs);
}
extern "C" int srsran_cbsegm_ldpc_bg2(srsran_cbsegm_t* s, uint32_t tbs)
{
  return cbsegm_ldpc(s, 1, tbs);
}

struct srsran_hip_sch_nr {
  DeviceTag tag;
  float    scaling = 0.8f;
  uint32_t max_iter = 10, max_cb = 0;
  srsran_hip_nr_sch_t*                         rm = nullptr;
  std::map<uint32_t, srsran_hip_ldpc_batch_t*> dec; // bg << 16 | Z
  uint8_t*  d_msg   = nullptr; // max_cb x MSG_STRIDE, one bit per byte
  uint8_t*  d_flags = nullptr; // max_cb
  uint8_t*  h_flags = nullptr; // pinned
  int *     d_iter = nullptr, *h_iter = nullptr;
  size_t    iter_cap = 0;
  uint32_t *d_map = nullptr, *h_map = nullptr;
  size_t    map_cap = 0;
  nrsch::CbFin *d_cbf = nullptr, *h_cbf = nullptr;
  size_t        cbf_cap = 0;
  nrsch::TbFin *d_tbf = nullptr, *h_tbf = nullptr;
  size_t        tbf_cap = 0;
  nrsch::TbFinRes *d_res = nullptr, *h_res = nullptr;
  size_t           res_cap = 0;
  // transport-block CRC: one row of 256 lane multipliers per (chunk size, CRC order) seen so far (nrsch::tb_finish_multipliers)
  std::map<uint32_t, uint32_t> crc_row; // chunk | order << 24 -> row
  uint32_t*                    d_crc_mult = nullptr;
  uint32_t                     crc_rows_cap = 0;
  // transmit side
  uint8_t*      d_cw = nullptr; // max_cb x CW_STRIDE code words (one bit per byte, filler marks kept)
  nrsch::TbEnc *d_tbe = nullptr, *h_tbe = nullptr;
  size_t        tbe_cap = 0;
  nrsch::CbEnc *d_cbe = nullptr, *h_cbe = nullptr;
  size_t        cbe_cap = 0;
  uint32_t*     d_tbcrc = nullptr;
  size_t        tbcrc_cap = 0;
};
static const uint32_t MSG_STRIDE = 8448;
static const uint32_t CW_STRIDE  = 66 * 384;

extern "C" int srsran_hip_sch_nr_create(srsran_hip_sch_nr_t** hh, float scaling_fctr, uint32_t max_nof_iter, uint32_t max_cb)
{
  if (!hh || max_cb == 0 || max_cb > 65535) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  auto* h     = new srsran_hip_sch_nr;
  h->scaling  = std::isnormal(scaling_fctr) ? scaling_fctr : 0.8f; // sch_nr.c:275
  h->max_iter = max_nof_iter ? max_nof_iter : 10;                  // ldpc_decoder.c:42,579
  h->max_cb   = max_cb;
  if (srsran_hip_nr_sch_create(&h->rm) != SRSRAN_SUCCESS || hipMalloc(&h->d_msg, (size_t)max_cb * MSG_STRIDE) != hipSuccess ||
      hipMalloc(&h->d_flags, max_cb) != hipSuccess || host_image_alloc(&h->h_flags, max_cb) != hipSuccess) {
    srsran_hip_sch_nr_free(h);
    return SRSRAN_ERROR;
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_sch_nr_free(srsran_hip_sch_nr_t* h)
{
  if (!h) {
    return;
  }
  srsran_hip_nr_sch_free(h->rm);
  for (auto& kv : h->dec) {
    srsran_hip_ldpc_batch_free(kv.second);
  }
  (void)hipFree(h->d_msg);
  (void)hipFree(h->d_flags);
  (void)hipHostFree(h->h_flags);
  (void)hipFree(h->d_iter), (void)hipHostFree(h->h_iter);
  (void)hipFree(h->d_map), (void)hipHostFree(h->h_map);
  (void)hipFree(h->d_cbf), (void)hipHostFree(h->h_cbf);
  (void)hipFree(h->d_tbf), (void)hipHostFree(h->h_tbf);
  (void)hipFree(h->d_res), (void)hipHostFree(h->h_res);
  (void)hipFree(h->d_crc_mult);
  (void)hipFree(h->d_cw);
  (void)hipFree(h->d_tbe), (void)hipHostFree(h->h_tbe);
  (void)hipFree(h->d_cbe), (void)hipHostFree(h->h_cbe);
  (void)hipFree(h->d_tbcrc);
  delete h;
}

// row of the CRC multiplier table for a transport block: computed and uploaded the first time its (chunk size, CRC order) is seen
static uint32_t crc_row_of(srsran_hip_sch_nr_t* h, uint32_t tbs_bits, uint32_t order)
{
  const uint32_t chunk = nrsch::tb_finish_chunk(tbs_bits), key = chunk | (order << 24);
  auto           it    = h->crc_row.find(key);
  if (it != h->crc_row.end()) {
    return it->second;
  }
  const uint32_t row = (uint32_t)h->crc_row.size();
  if (row >= h->crc_rows_cap) {
    const uint32_t cap = h->crc_rows_cap ? 2 * h->crc_rows_cap : 16;
    uint32_t*      nd  = nullptr;
    if (hipMalloc(&nd, (size_t)cap * 256 * sizeof(uint32_t)) != hipSuccess ||
        (h->d_crc_mult && (hipMemcpy(nd, h->d_crc_mult, (size_t)row * 256 * sizeof(uint32_t), hipMemcpyDeviceToDevice) != hipSuccess ||
                             hipDeviceSynchronize() != hipSuccess))) { // nothing may still read the old table when it is freed below
      (void)hipFree(nd);
      set_error("sch_nr decode: device allocation of the CRC multiplier table failed");
      return 0xffffffffu;
    }
    (void)hipFree(h->d_crc_mult);
    h->d_crc_mult   = nd;
    h->crc_rows_cap = cap;
  }
  uint32_t m[256];
  nrsch::tb_finish_multipliers(chunk, order, m);
  if (upload(h->d_crc_mult + (size_t)row * 256, m, sizeof(m)) != hipSuccess) {
    set_error("sch_nr decode: upload of the CRC multiplier table failed");
    return 0xffffffffu;
  }
  h->crc_row[key] = row;
  return row;
}

// (tail: device-to-host copies a caller wants queued behind the last kernel and in front of the call's one host wait)
struct TailCopy {
  void*       dst;
  const void* src;
  size_t      bytes;
};
static int sch_nr_decode(srsran_hip_sch_nr_t* h, const int8_t* d_e_bits, const srsran_hip_nr_tb_t* tbs, uint32_t n_tb, int8_t* d_softbuffer,
                         uint32_t sb_stride, uint8_t* cb_crc, uint8_t* d_cb_data, uint32_t data_stride, uint8_t* d_payload,
                         srsran_hip_nr_tb_result_t* res, void* stream, const TailCopy* tail, int n_tail)
{
  TraceRange trace_("srsran_hip_sch_nr_decode");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_sch_nr_decode", SRSRAN_ERROR);
  }
  if (h && n_tb == 0) {
    return SRSRAN_SUCCESS;
  }
  if (!h || !d_e_bits || !tbs || !d_softbuffer || !cb_crc || !d_cb_data || !d_payload || !res) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t        st = (hipStream_t)stream;
  std::vector<TbCfg> cfg(n_tb);
  std::vector<Job>   jobs;
  for (uint32_t t = 0; t < n_tb; t++) {
    TbCfg& c = cfg[t];
    if (!tb_cfg(tbs[t], &c)) {
      set_error("sch_nr: transport block %u: invalid size / modulation / layers", t);
      return SRSRAN_ERROR;
    }
    // soft-buffer protection (sch_nr.c:556-559) and the rows this library keeps per code block
    if (tbs[t].first_cb + c.C > h->max_cb || sb_stride < c.N || data_stride < (c.Kr + 7) / 8 || c.Kr > MSG_STRIDE) {
      set_error("sch_nr: transport block %u does not fit the soft buffer (%u code blocks of %u soft bits)", t, c.C, c.N);
      return SRSRAN_ERROR;
    }
    // element offsets travel as 32-bit values (job lists, code-word map): refuse what would wrap instead of writing into other rows
    uint64_t e_total = tbs[t].e_offset;
    for (uint32_t r = 0; r < c.C; r++) {
      e_total += get_E(c, r);
    }
    if (e_total > 0xffffffffull || (uint64_t)(tbs[t].first_cb + c.C) * sb_stride > 0xffffffffull) {
      set_error("sch_nr: transport block %u: input or soft-buffer offsets beyond 2^32 elements", t);
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    uint32_t in = tbs[t].e_offset;
    for (uint32_t r = 0; r < c.C; r++) {
      const uint32_t cb = tbs[t].first_cb + r;
      const uint32_t E  = get_E(c, r);
      if (cb_crc[cb]) {
        continue; // :584-588; the input pointer only advances for processed blocks (:665)
      }
      // n_llr of srsran_ldpc_rm_rx_c (ldpc_rm.c:704-705)
      static const uint32_t basek0[4][2] = {{0, 0}, {17, 13}, {33, 25}, {56, 43}};
      const uint32_t        Ncb = c.N <= c.Nref ? c.N : c.Nref;
      const uint32_t        k0  = c.N <= c.Nref ? c.Z * basek0[tbs[t].rv & 3u][c.bg] : c.Z * ((basek0[tbs[t].rv & 3u][c.bg] * c.Nref) / c.N);
      Job                   j;
      j.tb = t, j.cb = cb, j.E = E, j.in_off = in, j.n_llr = std::min(k0 + E, Ncb), j.cb_len = c.Kp - c.L_cb;
      j.bg = c.bg, j.Z = c.Z, j.rv = tbs[t].rv & 3u, j.mod = tbs[t].mod, j.Nref = c.Nref, j.F = c.F;
      j.fresh = (tbs[t].rv & SRSRAN_HIP_NR_TB_NEW_DATA) ? 1u : 0u;
      j.poly  = c.L_cb ? 0x1800063u : (c.L_tb == 24 ? 0x1864CFBu : 0x11021u); // :611-615
      j.order = c.L_cb ? 24 : c.L_tb;
      jobs.push_back(j);
      in += E;
    }
  }
  const uint32_t n_jobs = (uint32_t)jobs.size();
  // flags of every code block of these transport blocks -> device
  uint32_t cb_lo = 0xffffffffu, cb_hi = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    cb_lo = std::min(cb_lo, tbs[t].first_cb);
    cb_hi = std::max(cb_hi, tbs[t].first_cb + cfg[t].C);
  }
  for (uint32_t i = cb_lo; i < cb_hi; i++) {
    h->h_flags[i] = cb_crc[i] ? 1 : 0;
  }
  // A slot's worth of code blocks or less: the kernels read the job lists from, and write flags / iteration counts / results into, the PINNED HOST
  // images themselves (mapped into the device's address space): seven stream operations fewer per call, 6-9 us each whatever the size
  // (tools/probe/roundtrip_probe.hip).  Large batches keep the copies.
  const bool direct = n_jobs + n_tb <= 1024;
  uint8_t*   x_flags = direct ? h->h_flags : h->d_flags;
  if (!direct) {
    PHY_HIP_CHECK(hipMemcpyAsync(h->d_flags + cb_lo, h->h_flags + cb_lo, cb_hi - cb_lo, hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  }

  if (n_jobs) {
    // ---- rate de-matching, grouped by the parameters of init_rm
    std::vector<uint32_t> order(n_jobs);
    for (uint32_t i = 0; i < n_jobs; i++) {
      order[i] = i;
    }
    // (one integer key per job; a call whose blocks all share their parameters -- the usual case -- is already in order)
    auto rm_key = [&](const Job& j) {
      return ((uint64_t)j.bg << 60) | ((uint64_t)j.Z << 48) | ((uint64_t)j.rv << 46) | ((uint64_t)j.mod << 40) | ((uint64_t)j.Nref << 20) | (uint64_t)j.F;
    };
    auto by_rm = [&](uint32_t a, uint32_t b) { return rm_key(jobs[a]) < rm_key(jobs[b]); };
    if (!std::is_sorted(order.begin(), order.end(), by_rm)) {
      std::stable_sort(order.begin(), order.end(), by_rm);
    }
    std::vector<srsran_hip_ldpc_cb_t> cbs;
    std::vector<uint8_t>              fresh;
    for (uint32_t i = 0; i < n_jobs;) {
      uint32_t e = i;
      cbs.clear();
      fresh.clear();
      while (e < n_jobs && rm_key(jobs[order[e]]) == rm_key(jobs[order[i]])) {
        const Job& j = jobs[order[e]];
        cbs.push_back(srsran_hip_ldpc_cb_t{j.in_off, j.cb * sb_stride, j.E});
        fresh.push_back((uint8_t)j.fresh);
        e++;
      }
      const Job& j = jobs[order[i]];
      if (srsran_hip_ldpc_rm_rx_batch_new(h->rm, SRSRAN_HIP_LLR_BYTE, d_e_bits, d_softbuffer, cbs.data(), fresh.data(), (uint32_t)cbs.size(), j.F,
                                          (srsran_basegraph_t)j.bg, j.Z, j.rv, (srsran_mod_t)j.mod, j.Nref, st) != SRSRAN_SUCCESS) {
        set_error("sch_nr: rate de-matching refused (E=%u, Z=%u)", j.E, j.Z);
        return SRSRAN_ERROR;
      }
      i = e;
    }
    // ---- decoding with CRC early stop, grouped by decoder and code-word length
    auto dec_key = [&](const Job& j) { return ((uint64_t)j.bg << 60) | ((uint64_t)j.Z << 48) | ((uint64_t)j.n_llr << 28) | (uint64_t)(j.poly & 0xfffffffu); };
    auto by_dec  = [&](uint32_t a, uint32_t b) { return dec_key(jobs[a]) < dec_key(jobs[b]); };
    if (!std::is_sorted(order.begin(), order.end(), by_dec)) {
      std::stable_sort(order.begin(), order.end(), by_dec);
    }
    if (!ensure(&h->d_iter, &h->h_iter, &h->iter_cap, n_jobs) || !ensure(&h->d_map, &h->h_map, &h->map_cap, n_jobs) ||
        !ensure(&h->d_cbf, &h->h_cbf, &h->cbf_cap, n_jobs)) {
      return SRSRAN_ERROR;
    }
    for (uint32_t i = 0; i < n_jobs; i++) {
      const Job& j = jobs[order[i]];
      h->h_map[i]  = j.cb;
      h->h_cbf[i]  = nrsch::CbFin{j.cb, j.cb, j.cb_len};
    }
    uint32_t*     x_map  = direct ? h->h_map : h->d_map;
    nrsch::CbFin* x_cbf  = direct ? h->h_cbf : h->d_cbf;
    int*          x_iter = direct ? h->h_iter : h->d_iter;
    if (!direct) {
      PHY_HIP_CHECK(hipMemcpyAsync(h->d_map, h->h_map, n_jobs * sizeof(uint32_t), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
      PHY_HIP_CHECK(hipMemcpyAsync(h->d_cbf, h->h_cbf, n_jobs * sizeof(nrsch::CbFin), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    }
    for (uint32_t i = 0; i < n_jobs;) {
      uint32_t e = i;
      while (e < n_jobs && dec_key(jobs[order[e]]) == dec_key(jobs[order[i]])) {
        e++;
      }
      const Job&     j   = jobs[order[i]];
      const uint32_t key = ((uint32_t)j.bg << 16) | j.Z;
      auto           it  = h->dec.find(key);
      if (it == h->dec.end()) {
        srsran_hip_ldpc_batch_t* d = nullptr;
        if (srsran_hip_ldpc_batch_create(&d, (srsran_basegraph_t)j.bg, (uint16_t)j.Z, h->scaling, h->max_iter, h->max_cb) != SRSRAN_SUCCESS) {
          return SRSRAN_ERROR;
        }
        it = h->dec.emplace(key, d).first;
      }
      if (srsran_hip_ldpc_batch_run_crc_map(it->second, d_softbuffer, sb_stride, h->d_msg, MSG_STRIDE, x_map + i, e - i, j.n_llr, j.poly, j.order,
                                            x_iter + i, st) != SRSRAN_SUCCESS) {
        return SRSRAN_ERROR;
      }
      i = e;
    }
    PHY_HIP_CHECK(nrsch::launch_cb_finish(h->d_msg, MSG_STRIDE, x_cbf, x_iter, n_jobs, x_flags, d_cb_data, data_stride, st), SRSRAN_ERROR);
    if (!direct) {
      PHY_HIP_CHECK(hipMemcpyAsync(h->h_iter, h->d_iter, n_jobs * sizeof(int), hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    }
    // ---- back to transport blocks
    if (!ensure(&h->d_tbf, &h->h_tbf, &h->tbf_cap, n_tb) || !ensure(&h->d_res, &h->h_res, &h->res_cap, n_tb)) {
      return SRSRAN_ERROR;
    }
    for (uint32_t t = 0; t < n_tb; t++) {
      h->h_tbf[t] = nrsch::TbFin{tbs[t].first_cb, cfg[t].C, cfg[t].Kp, cfg[t].L_cb, cfg[t].L_tb, cfg[t].A, tbs[t].payload_offset,
                                 crc_row_of(h, cfg[t].A, cfg[t].L_tb)};
      if (h->h_tbf[t].mult == 0xffffffffu) {
        return SRSRAN_ERROR;
      }
    }
    if (!direct) {
      PHY_HIP_CHECK(hipMemcpyAsync(h->d_tbf, h->h_tbf, n_tb * sizeof(nrsch::TbFin), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    }
    PHY_HIP_CHECK(nrsch::launch_tb_finish(d_cb_data, data_stride, x_flags, direct ? h->h_tbf : h->d_tbf, n_tb, d_payload, h->d_crc_mult,
                                          direct ? h->h_res : h->d_res, st), SRSRAN_ERROR);
    if (!direct) {
      PHY_HIP_CHECK(hipMemcpyAsync(h->h_res, h->d_res, n_tb * sizeof(nrsch::TbFinRes), hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
      PHY_HIP_CHECK(hipMemcpyAsync(h->h_flags + cb_lo, h->d_flags + cb_lo, cb_hi - cb_lo, hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    }
    for (int i = 0; i < n_tail; i++) {
    if (tail[i].bytes) {
      PHY_HIP_CHECK(hipMemcpyAsync(tail[i].dst, tail[i].src, tail[i].bytes, hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    }
  }
  PHY_HIP_CHECK(hipStreamSynchronize(st), SRSRAN_ERROR);
    std::vector<uint32_t> it_sum(n_tb, 0);
    for (uint32_t i = 0; i < n_jobs; i++) {
      const Job& j = jobs[order[i]];
      it_sum[j.tb] += h->h_iter[i] == 0 ? h->max_iter : (uint32_t)h->h_iter[i]; // :629-631
    }
    for (uint32_t t = 0; t < n_tb; t++) {
      for (uint32_t r = 0; r < cfg[t].C; r++) {
        cb_crc[tbs[t].first_cb + r] = h->h_flags[tbs[t].first_cb + r];
      }
      res[t].all_decoded = h->h_res[t].all_decoded;
      res[t].crc_ok      = h->h_res[t].crc_ok;
      res[t].avg_iter    = (float)it_sum[t] / (float)cfg[t].C; // :657-661
      res[t].nof_cb      = cfg[t].C;
    }
    return SRSRAN_SUCCESS;
  }
  // nothing left to decode (every code block already had its CRC): assembly and transport CRC only
  if (!ensure(&h->d_tbf, &h->h_tbf, &h->tbf_cap, n_tb) || !ensure(&h->d_res, &h->h_res, &h->res_cap, n_tb)) {
    return SRSRAN_ERROR;
  }
  for (uint32_t t = 0; t < n_tb; t++) {
    h->h_tbf[t] = nrsch::TbFin{tbs[t].first_cb, cfg[t].C, cfg[t].Kp, cfg[t].L_cb, cfg[t].L_tb, cfg[t].A, tbs[t].payload_offset,
                               crc_row_of(h, cfg[t].A, cfg[t].L_tb)};
    if (h->h_tbf[t].mult == 0xffffffffu) {
      return SRSRAN_ERROR;
    }
  }
  if (!direct) {
    PHY_HIP_CHECK(hipMemcpyAsync(h->d_tbf, h->h_tbf, n_tb * sizeof(nrsch::TbFin), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  }
  PHY_HIP_CHECK(nrsch::launch_tb_finish(d_cb_data, data_stride, x_flags, direct ? h->h_tbf : h->d_tbf, n_tb, d_payload, h->d_crc_mult,
                                        direct ? h->h_res : h->d_res, st), SRSRAN_ERROR);
  if (!direct) {
    PHY_HIP_CHECK(hipMemcpyAsync(h->h_res, h->d_res, n_tb * sizeof(nrsch::TbFinRes), hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
  }
  for (int i = 0; i < n_tail; i++) {
    if (tail[i].bytes) {
      PHY_HIP_CHECK(hipMemcpyAsync(tail[i].dst, tail[i].src, tail[i].bytes, hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    }
  }
  PHY_HIP_CHECK(hipStreamSynchronize(st), SRSRAN_ERROR);
  for (uint32_t t = 0; t < n_tb; t++) {
    res[t].all_decoded = h->h_res[t].all_decoded;
    res[t].crc_ok      = h->h_res[t].crc_ok;
    res[t].avg_iter    = 0.0f;
    res[t].nof_cb      = cfg[t].C;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_sch_nr_decode(srsran_hip_sch_nr_t* h, const int8_t* d_e_bits, const srsran_hip_nr_tb_t* tbs, uint32_t n_tb,
                                        int8_t* d_softbuffer, uint32_t sb_stride, uint8_t* cb_crc, uint8_t* d_cb_data, uint32_t data_stride,
                                        uint8_t* d_payload, srsran_hip_nr_tb_result_t* res, void* stream)
{
  return sch_nr_decode(h, d_e_bits, tbs, n_tb, d_softbuffer, sb_stride, cb_crc, d_cb_data, data_stride, d_payload, res, stream, nullptr, 0);
}

// ------------------------------------------------------------------------------------------------ the reference's transport-block entry points
//
// sch_nr_decode (sch_nr.c:522-713) as srsran_dlsch_nr_decode / srsran_ulsch_nr_decode (:724-749) reach it, for ONE transport block on the
// caller's HOST buffers in the reference's soft-buffer struct: one upload, one de-matching launch, one early-stop LDPC launch, the
// code-block and transport-block finish kernels, one download.  The staging context is private to the calling thread.

namespace {

struct NrTbStage {
  hipStream_t                               st = nullptr;
  std::map<uint64_t, srsran_hip_sch_nr_t*>  sch; // (scaling factor, iterations) -> decoder object
  uint8_t*                                  pin = nullptr; // pinned image: [soft rows | data rows | e bits | payload]
  uint8_t*                                  dev = nullptr;
  size_t                                    cap = 0;
  bool                                      tried = false;
  static const uint32_t                     MAX_CB = 160; // > SRSRAN_SCH_NR_MAX_NOF_CB_LDPC (sch_nr.h:41)
  ~NrTbStage()
  {
    for (auto& kv : sch) {
      srsran_hip_sch_nr_free(kv.second);
    }
    (void)hipFree(dev);
    (void)hipHostFree(pin);
    if (st) {
      (void)hipStreamDestroy(st);
    }
  }
  bool ready()
  {
    if (!tried) {
      tried = true;
      if (device_available()) {
        bind_thread();
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
          st = nullptr;
        }
      }
    }
    return st != nullptr;
  }
  srsran_hip_sch_nr_t* decoder(float scaling, uint32_t iters)
  {
    uint32_t sbits;
    memcpy(&sbits, &scaling, 4);
    const uint64_t key = ((uint64_t)sbits << 32) | iters;
    auto           it  = sch.find(key);
    if (it != sch.end()) {
      return it->second;
    }
    srsran_hip_sch_nr_t* h = nullptr;
    if (srsran_hip_sch_nr_create(&h, scaling, iters, MAX_CB) != SRSRAN_SUCCESS) {
      return nullptr;
    }
    sch[key] = h;
    return h;
  }
  bool grow(size_t need)
  {
    if (need <= cap) {
      return true;
    }
    (void)hipFree(dev);
    (void)hipHostFree(pin);
    dev = pin = nullptr;
    cap = 0;
    if (hipMalloc((void**)&dev, need) != hipSuccess || host_image_alloc(&pin, need) != hipSuccess) {
      return false;
    }
    cap = need;
    return true;
  }
};

// true when the n bytes at p (8-byte aligned, as the rows of a soft buffer are) are all zero: a row straight after srsran_softbuffer_rx_reset
// is not worth a copy, let alone a transfer.  Read-only, four independent accumulators (vectorises), early exit per 4 KB.
inline bool all_zero(const uint8_t* p, size_t n)
{
  size_t i = 0;
  if ((reinterpret_cast<uintptr_t>(p) & 7u) == 0) {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    const size_t    w = n / 8;
    for (size_t j = 0; j < w;) {
      const size_t e = j + 512 < w ? j + 512 : w;
      uint64_t     a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (; j + 4 <= e; j += 4) {
        a0 |= q[j], a1 |= q[j + 1], a2 |= q[j + 2], a3 |= q[j + 3];
      }
      for (; j < e; j++) {
        a0 |= q[j];
      }
      if (a0 | a1 | a2 | a3) {
        return false;
      }
    }
    i = w * 8;
  }
  for (; i < n; i++) {
    if (p[i]) {
      return false;
    }
  }
  return true;
}

} // namespace

extern "C" int srsran_hip_sch_nr_decode_tb(float scaling_fctr, uint32_t max_nof_iter, const srsran_hip_nr_tb_t* tb_in, const int8_t* e_bits,
                                           srsran_softbuffer_rx_t* softbuffer, uint8_t* payload, bool* crc, float* avg_iter)
{
  if (!tb_in || !e_bits || !softbuffer || !payload || !crc || !avg_iter || !softbuffer->buffer_f || !softbuffer->data || !softbuffer->cb_crc) {
    return SRSRAN_ERROR_INVALID_INPUTS; // sch_nr.c:528-531
  }
  TbCfg c;
  if (!tb_cfg(*tb_in, &c)) {
    fprintf(stderr, "[srsran_phy_hip] sch_nr decode: invalid transport block (tbs %u, mod %u, layers %u)\n", tb_in->tbs, tb_in->mod, tb_in->N_L);
    return SRSRAN_ERROR;
  }
  if (softbuffer->max_cb < c.C || softbuffer->max_cb_size < c.N || c.C > NrTbStage::MAX_CB) { // :556-559
    return SRSRAN_ERROR;
  }
  NrTbStage& s = thread_device_local<NrTbStage>();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] sch_nr decode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  srsran_hip_sch_nr_t* h = s.decoder(scaling_fctr, max_nof_iter);
  if (!h) {
    return SRSRAN_ERROR;
  }
  auto           al         = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const uint32_t sb_stride  = (uint32_t)al(c.N);
  const uint32_t data_stride = (uint32_t)al((c.Kr + 7) / 8);
  const uint32_t cb_bytes   = (c.Kp - c.L_cb + 7) / 8; // packed bits of a decoded code block, softbuffer.rx->data[r] (:650-652)
  uint8_t        flags[NrTbStage::MAX_CB];
  size_t         n_e = 0;
  for (uint32_t r = 0; r < c.C; r++) {
    flags[r] = softbuffer->cb_crc[r] ? 1 : 0;
    if (!flags[r]) {
      n_e += get_E(c, r); // the input holds the soft bits of the still undecoded blocks, back to back (:665)
    }
    if (!softbuffer->buffer_f[r] || !softbuffer->data[r]) {
      fprintf(stderr, "Error: soft-buffer provided NULL buffer for cb_idx=%u\n", r); // :571-574
      return SRSRAN_ERROR;
    }
  }
  const size_t o_soft = 0, o_data = al(o_soft + (size_t)c.C * sb_stride), o_e = al(o_data + (size_t)c.C * data_stride), o_pay = al(o_e + n_e);
  if (!s.grow(al(o_pay + c.A / 8 + 8))) {
    fprintf(stderr, "[srsran_phy_hip] sch_nr decode: staging allocation failed\n");
    return SRSRAN_ERROR;
  }
  bool any_flag = false, any_soft = false;
  int  first = -1, last = -1;
  for (uint32_t r = 0; r < c.C; r++) {
    if (flags[r]) {
      any_flag = true;
      memcpy(s.pin + o_data + (size_t)r * data_stride, softbuffer->data[r], cb_bytes); // decoded earlier: its packed bits join the assembly
    } else {
      any_soft = any_soft || !all_zero(reinterpret_cast<const uint8_t*>(softbuffer->buffer_f[r]), c.N);
      first    = first < 0 ? (int)r : first;
      last     = (int)r;
    }
  }
  if (any_soft || any_flag) { // a retransmission: the rows hold the earlier transmissions' soft bits
    for (uint32_t r = 0; r < c.C; r++) {
      if (!flags[r]) {
        memcpy(s.pin + o_soft + (size_t)r * sb_stride, softbuffer->buffer_f[r], c.N);
      }
    }
  }
  srsran_hip_nr_tb_t tb = *tb_in;
  tb.rv &= 3u;
  tb.e_offset = tb.payload_offset = tb.first_cb = 0;
  if (first >= 0) {
    memcpy(s.pin + o_e, e_bits, n_e); // (the de-matcher reads them once, coalesced: straight from the pinned image)
    if (!any_soft && !any_flag) {
      tb.rv |= SRSRAN_HIP_NR_TB_NEW_DATA; // rows as srsran_softbuffer_rx_reset left them: written, not accumulated into, and not uploaded
    } else {
      PHY_HIP_CHECK(hipMemcpyAsync(s.dev + o_soft + (size_t)first * sb_stride, s.pin + o_soft + (size_t)first * sb_stride,
                                   (size_t)(last - first) * sb_stride + c.N, hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
    }
  }
  if (any_flag) {
    PHY_HIP_CHECK(hipMemcpyAsync(s.dev + o_data, s.pin + o_data, (size_t)c.C * data_stride, hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  }
  srsran_hip_nr_tb_result_t res = {};
  const TailCopy tail[2] = {{s.pin + o_data, s.dev + o_data, (size_t)c.C * data_stride}, {s.pin + o_pay, s.dev + o_pay, c.A / 8}};
  if (sch_nr_decode(h, reinterpret_cast<const int8_t*>(s.pin + o_e), &tb, 1, reinterpret_cast<int8_t*>(s.dev + o_soft), sb_stride, flags, s.dev + o_data,
                    data_stride, s.dev + o_pay, &res, s.st, tail, 2) != SRSRAN_SUCCESS) {
    (void)hipStreamSynchronize(s.st); // nothing of a failed call may still be in flight when the next one re-uses the images
    fprintf(stderr, "[srsran_phy_hip] sch_nr decode: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  // host side effects of :633-652: flags, the packed bits of the blocks decoded now; rows of the blocks that are still undecoded
  int f_first = -1, f_last = -1;
  for (uint32_t r = 0; r < c.C; r++) {
    if (softbuffer->cb_crc[r]) {
      continue;
    }
    if (flags[r]) {
      softbuffer->cb_crc[r] = true;
      memcpy(softbuffer->data[r], s.pin + o_data + (size_t)r * data_stride, cb_bytes);
    } else {
      f_first = f_first < 0 ? (int)r : f_first;
      f_last  = (int)r;
    }
  }
  if (f_first >= 0) {
    PHY_HIP_CHECK(hipMemcpyAsync(s.pin + o_soft + (size_t)f_first * sb_stride, s.dev + o_soft + (size_t)f_first * sb_stride,
                                 (size_t)(f_last - f_first) * sb_stride + c.N, hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
    // (the circular buffer of a block ends at Ncb = min(N, Nref), ldpc_rm.c:704-705: nothing behind it is ever written or read)
    const uint32_t n_cb_buf = c.N <= c.Nref ? c.N : c.Nref;
    for (int r = f_first; r <= f_last; r++) {
      if (!softbuffer->cb_crc[r]) {
        memcpy(softbuffer->buffer_f[r], s.pin + o_soft + (size_t)r * sb_stride, n_cb_buf);
      }
    }
  }
  *avg_iter = res.avg_iter; // :657-661
  if (res.all_decoded) {    // :664-666: otherwise neither the payload nor res->crc is touched
    memcpy(payload, s.pin + o_pay, c.A / 8);
    *crc = res.crc_ok != 0;
  }
  return SRSRAN_SUCCESS;
}

// sch_nr_encode (sch_nr.c:375-520) for a batch: payload bytes -> rate-matched bits (one per byte) of every code block, back to back per
// transport block from e_offset on.  Asynchronous on `stream`.
extern "C" int srsran_hip_sch_nr_encode(srsran_hip_sch_nr_t* h, const uint8_t* d_payload, const srsran_hip_nr_tb_t* tbs, uint32_t n_tb, uint8_t* d_e_bits,
                                        void* stream)
{
  TraceRange trace_("srsran_hip_sch_nr_encode");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_sch_nr_encode", SRSRAN_ERROR);
  }
  if (h && n_tb == 0) {
    return SRSRAN_SUCCESS;
  }
  if (!h || !d_payload || !tbs || !d_e_bits) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t st = (hipStream_t)stream;
  struct EJob {
    int      bg;
    uint32_t Z, rv, mod, Nref, E, out_off, row;
  };
  std::vector<TbCfg> cfg(n_tb);
  std::vector<EJob>  jobs;
  if (!ensure(&h->d_tbe, &h->h_tbe, &h->tbe_cap, n_tb)) {
    return SRSRAN_ERROR;
  }
  uint32_t n_cb = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    if (!tb_cfg(tbs[t], &cfg[t])) {
      set_error("sch_nr: transport block %u: invalid size / modulation / layers", t);
      return SRSRAN_ERROR;
    }
    n_cb += cfg[t].C;
  }
  if (n_cb > h->max_cb) {
    set_error("sch_nr: %u code blocks in one call, the object was created for %u", n_cb, h->max_cb);
    return SRSRAN_ERROR;
  }
  if (!ensure(&h->d_cbe, &h->h_cbe, &h->cbe_cap, n_cb)) {
    return SRSRAN_ERROR;
  }
  if (!h->d_cw) {
    PHY_HIP_CHECK(hipMalloc(&h->d_cw, (size_t)h->max_cb * CW_STRIDE), SRSRAN_ERROR);
  }
  if (n_tb > h->tbcrc_cap) {
    (void)hipFree(h->d_tbcrc);
    h->d_tbcrc   = nullptr;
    h->tbcrc_cap = 0;
    PHY_HIP_CHECK(hipMalloc(&h->d_tbcrc, ((size_t)n_tb + 16) * sizeof(uint32_t)), SRSRAN_ERROR);
    h->tbcrc_cap = (size_t)n_tb + 16;
  }
  uint32_t row = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    const TbCfg& c = cfg[t];
    h->h_tbe[t]    = nrsch::TbEnc{tbs[t].payload_offset, c.A, c.L_tb};
    uint32_t bit = 0, out = tbs[t].e_offset;
    for (uint32_t r = 0; r < c.C; r++) {
      const bool     last   = r == c.C - 1;
      const uint32_t cb_len = c.Kp - c.L_cb - (last ? c.L_tb : 0);
      h->h_cbe[row]         = nrsch::CbEnc{t, bit, cb_len, c.Kp, c.Kr, c.L_cb, last ? 1u : 0u, row};
      bit += (cb_len / 8) * 8; // input_ptr += cb_len / 8 (:448)
      const uint32_t E = get_E(c, r);
      jobs.push_back(EJob{c.bg, c.Z, tbs[t].rv & 3u, tbs[t].mod, c.Nref, E, out, row});
      out += E;
      row++;
    }
  }
  PHY_HIP_CHECK(hipMemcpyAsync(h->d_tbe, h->h_tbe, n_tb * sizeof(nrsch::TbEnc), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(h->d_cbe, h->h_cbe, n_cb * sizeof(nrsch::CbEnc), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  PHY_HIP_CHECK(nrsch::launch_tb_crc_enc(d_payload, h->d_tbe, n_tb, h->d_tbcrc, st), SRSRAN_ERROR);
  PHY_HIP_CHECK(nrsch::launch_cb_build(d_payload, h->d_cbe, n_cb, h->d_tbe, h->d_tbcrc, h->d_msg, MSG_STRIDE, st), SRSRAN_ERROR);
  // LDPC encoder per (graph, Z), rate matcher per (graph, Z, rv, modulation, Nref)
  std::vector<uint32_t> order(n_cb);
  for (uint32_t i = 0; i < n_cb; i++) {
    order[i] = i;
  }
  auto key = [&](const EJob& j) { return std::make_tuple(j.bg, j.Z, j.rv, j.mod, j.Nref); };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(jobs[a]) < key(jobs[b]); });
  std::vector<srsran_hip_ldpc_cb_t> cbs;
  for (uint32_t i = 0; i < n_cb;) {
    uint32_t e = i;
    while (e < n_cb && jobs[order[e]].bg == jobs[order[i]].bg && jobs[order[e]].Z == jobs[order[i]].Z) {
      e++;
    }
    const EJob&    j = jobs[order[i]];
    const uint32_t N = j.Z * (j.bg == 0 ? 66u : 50u);
    cbs.clear();
    for (uint32_t k = i; k < e; k++) {
      cbs.push_back(srsran_hip_ldpc_cb_t{jobs[order[k]].row * MSG_STRIDE, jobs[order[k]].row * CW_STRIDE, N});
    }
    if (srsran_hip_ldpc_encode_batch(h->rm, h->d_msg, h->d_cw, cbs.data(), (uint32_t)cbs.size(), (srsran_basegraph_t)j.bg, j.Z, st) != SRSRAN_SUCCESS) {
      return SRSRAN_ERROR;
    }
    i = e;
  }
  for (uint32_t i = 0; i < n_cb;) {
    uint32_t e = i;
    cbs.clear();
    while (e < n_cb && key(jobs[order[e]]) == key(jobs[order[i]])) {
      cbs.push_back(srsran_hip_ldpc_cb_t{jobs[order[e]].row * CW_STRIDE, jobs[order[e]].out_off, jobs[order[e]].E});
      e++;
    }
    const EJob& j = jobs[order[i]];
    if (srsran_hip_ldpc_rm_tx_batch(h->rm, h->d_cw, d_e_bits, cbs.data(), (uint32_t)cbs.size(), (srsran_basegraph_t)j.bg, j.Z, j.rv, (srsran_mod_t)j.mod,
                                    j.Nref, st) != SRSRAN_SUCCESS) {
      return SRSRAN_ERROR;
    }
    i = e;
  }
  return SRSRAN_SUCCESS;
}

// sch_nr_encode (sch_nr.c:375-520) as srsran_dlsch_nr_encode / srsran_ulsch_nr_encode (:715-741) reach it, for ONE transport block on the caller's
// HOST buffers: payload bytes in, rate-matched bits (one per byte) of every code block out, back to back.  Stateless: the code words the
// reference parks in softbuffer.tx->buffer_b[r] (it re-encodes on every call as well, :434-470) are not written.
extern "C" int srsran_hip_sch_nr_encode_tb(const srsran_hip_nr_tb_t* tb_in, const uint8_t* data, uint8_t* e_bits)
{
  if (!tb_in || !data || !e_bits) {
    return SRSRAN_ERROR_INVALID_INPUTS; // sch_nr.c:381-383
  }
  TbCfg c;
  if (!tb_cfg(*tb_in, &c) || c.C > NrTbStage::MAX_CB) {
    fprintf(stderr, "[srsran_phy_hip] sch_nr encode: invalid transport block (tbs %u, mod %u, layers %u)\n", tb_in->tbs, tb_in->mod, tb_in->N_L);
    return SRSRAN_ERROR;
  }
  NrTbStage& s = thread_device_local<NrTbStage>();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] sch_nr encode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  srsran_hip_sch_nr_t* h = s.decoder(0.8f, 10); // (the transmit side has no decoder parameters: one object per thread)
  if (!h) {
    return SRSRAN_ERROR;
  }
  size_t n_e = 0;
  for (uint32_t r = 0; r < c.C; r++) {
    n_e += get_E(c, r);
  }
  auto         al    = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_pay = 0, o_e = al(o_pay + c.A / 8 + 8);
  if (!s.grow(al(o_e + n_e))) {
    fprintf(stderr, "[srsran_phy_hip] sch_nr encode: staging allocation failed\n");
    return SRSRAN_ERROR;
  }
  memcpy(s.pin + o_pay, data, c.A / 8);
  srsran_hip_nr_tb_t tb = *tb_in;
  tb.rv &= 3u;
  tb.e_offset = tb.payload_offset = tb.first_cb = 0;
  // the kernels read the payload from, and the rate matcher writes the bits into, the pinned host image itself (read twice, written once)
  if (srsran_hip_sch_nr_encode(h, s.pin + o_pay, &tb, 1, s.pin + o_e, s.st) != SRSRAN_SUCCESS) {
    (void)hipStreamSynchronize(s.st);
    fprintf(stderr, "[srsran_phy_hip] sch_nr encode: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  memcpy(e_bits, s.pin + o_e, n_e);
  return SRSRAN_SUCCESS;
}
