// modem_host.cpp -- C ABI of the soft demodulator / descrambler (include/srsran_amd/phy_modem_abi.h).
#include "hip_common.h"
#include "modem_device.h"
#include "srsran_amd/phy_modem_abi.h"

#include <cmath>
#include <mutex>
#include <vector>

using namespace phyhip;

namespace {

// ---- Gold sequence jump tables (TS 36.211 7.2: x1(n+31) = x1(n+3)+x1(n), x2(n+31) = x2(n+3)+x2(n+2)+x2(n+1)+x2(n)) -------
inline uint32_t adv16_x1(uint32_t s)
{
  return (s >> 16) | ((((s >> 3) ^ s) & 0xffffu) << 15);
}
inline uint32_t adv16_x2(uint32_t s)
{
  return (s >> 16) | ((((s >> 3) ^ (s >> 2) ^ (s >> 1) ^ s) & 0xffffu) << 15);
}

struct SeqTables {
  std::mutex mu;
  uint32_t*  d_x1   = nullptr;
  uint32_t*  d_x2   = nullptr;
  bool       failed = false;
};
#define g_seq (device_local<SeqTables>()) // the sequence tables of the calling thread's device

bool seq_tables(const uint32_t** x1, const uint32_t** x2)
{
  std::lock_guard<std::mutex> lk(g_seq.mu);
  if (!g_seq.d_x1 && !g_seq.failed) {
    // x2: register after Nc = 1600 chips, then every 128 chips, as 31 columns (the map seed -> register is linear over
    // GF(2)); x1: the chips themselves, 32 per word
    std::vector<uint32_t> x1((size_t)MODEM_SEQ_NCHUNKS * (MODEM_SEQ_CHUNK / 32)), x2((size_t)MODEM_SEQ_NCHUNKS * 31);
    uint32_t              s1 = 1, col[31];
    for (int i = 0; i < 31; i++) {
      col[i] = 1u << i;
    }
    for (int n = 0; n < 1600 / 16; n++) {
      s1 = adv16_x1(s1);
      for (int i = 0; i < 31; i++) {
        col[i] = adv16_x2(col[i]);
      }
    }
    for (uint32_t j = 0; j < MODEM_SEQ_NCHUNKS; j++) {
      for (int i = 0; i < 31; i++) {
        x2[(size_t)j * 31 + i] = col[i];
      }
      for (uint32_t n = 0; n < MODEM_SEQ_CHUNK / 16; n++) {
        x1[(size_t)j * (MODEM_SEQ_CHUNK / 32) + n / 2] |= (s1 & 0xffffu) << (16 * (n & 1));
        s1 = adv16_x1(s1);
        for (int i = 0; i < 31; i++) {
          col[i] = adv16_x2(col[i]);
        }
      }
    }
    if (hipMalloc(&g_seq.d_x1, x1.size() * 4) != hipSuccess || hipMalloc(&g_seq.d_x2, x2.size() * 4) != hipSuccess ||
        upload(g_seq.d_x1, x1.data(), x1.size() * 4) != hipSuccess ||
        upload(g_seq.d_x2, x2.data(), x2.size() * 4) != hipSuccess) {
      set_error("modem: cannot allocate the sequence tables on the device");
      g_seq.failed = true;
    }
  }
  *x1 = g_seq.d_x1;
  *x2 = g_seq.d_x2;
  return !g_seq.failed;
}

modem::Consts make_consts()
{
  // the expressions of demod_soft.c, evaluated by the host compiler with the reference's types
  modem::Consts k;
  k.t16_tail_s = 2 * 400 / sqrtf(10);
  k.t16_tail_b = 2 * 30 / sqrtf(10);
  k.f16        = 2 / sqrtf(10);
  k.f64a       = 4 / sqrtf(42);
  k.f64b       = 2 / sqrtf(42);
  k.c8         = 8.0f / sqrtf(170.0f);
  k.c4         = 4.0f / sqrtf(170.0f);
  k.c2         = 2.0f / sqrtf(170.0f);
  k.qpsk_s     = (float)(-100 * M_SQRT2);
  k.qpsk_b     = (float)(-20 * M_SQRT2);
  k.qpsk_f     = (float)(-M_SQRT2);
  k.o16_s      = (int16_t)(2 * 400 / sqrtf(10));
  k.o64a_s     = (int16_t)(4 * 700 / sqrtf(42));
  k.o64b_s     = (int16_t)(2 * 700 / sqrtf(42));
  k.o16_b      = (int8_t)(2 * 30 / sqrtf(10));
  k.o64a_b     = (int8_t)(4 * 40 / sqrtf(42));
  k.o64b_b     = (int8_t)(2 * 40 / sqrtf(42));
  return k;
}

inline uint32_t bits_per_symbol(uint32_t mod)
{
  return mod == 0 ? 1 : (mod == modem::MOD_PASS ? 1 : 2 * mod);
}
inline size_t llr_size(int t)
{
  return t == modem::LLR_I16 ? 2 : (t == modem::LLR_I8 ? 1 : 4);
}

bool fill_params(modem::Params& p, int llr_type)
{
  memset(&p, 0, sizeof(p));
  p.llr_type = llr_type;
  p.k        = make_consts();
  return seq_tables(&p.x1_bits, &p.x2_cols);
}

// ---- constellation tables of TS 36.211 7.1.1-7.1.5 as lib/src/phy/modem/lte_tables.c:30-181 evaluates them (same float expressions on the host
// compiler: levels k / sqrtf(10 | 42 | 170), (float)M_SQRT1_2), index = the symbol's bits, first bit most significant
struct ModTables {
  std::mutex mu;
  float2*    d      = nullptr;
  bool       failed = false;
};
#define g_mod (device_local<ModTables>())

void build_mod_tables(std::vector<float2>& t)
{
  t.assign(2 + 4 + 16 + 64 + 256, make_float2(0.f, 0.f));
  const float l2 = (float)M_SQRT1_2;
  t[0] = make_float2(l2, l2); // BPSK: 0 -> (1 + j) / sqrt 2
  t[1] = make_float2(-l2, -l2);
  for (int i = 0; i < 4; i++) { // QPSK: b0 -> sign of I, b1 -> sign of Q
    t[2 + i] = make_float2((i & 2) ? -l2 : l2, (i & 1) ? -l2 : l2);
  }
  const float q16[2] = {1.0f / sqrtf(10.0f), 3.0f / sqrtf(10.0f)};
  for (int i = 0; i < 16; i++) { // b0 b1 signs, b2 b3 outer level
    const float re = q16[(i >> 1) & 1], im = q16[i & 1];
    t[6 + i]       = make_float2((i & 8) ? -re : re, (i & 4) ? -im : im);
  }
  const float q64[4] = {3.0f / sqrtf(42.0f), 1.0f / sqrtf(42.0f), 5.0f / sqrtf(42.0f), 7.0f / sqrtf(42.0f)}; // (b2 b4) / (b3 b5) = 00 01 10 11
  for (int i = 0; i < 64; i++) {
    const float re = q64[((i >> 3) & 1) * 2 + ((i >> 1) & 1)], im = q64[((i >> 2) & 1) * 2 + (i & 1)];
    t[22 + i]      = make_float2((i & 32) ? -re : re, (i & 16) ? -im : im);
  }
  for (int i = 0; i < 256; i++) { // lte_tables.c:162-181: the nested form (1 - 2 b0) [8 - (1 - 2 b2) [4 - (1 - 2 b4) [2 - (1 - 2 b6)]]] built from the inside
    float off = -1, re = 0, im = 0;
    for (int j = 0; j < 4; j++) {
      re += off;
      im += off;
      off *= 2;
      re *= (i & (1 << (2 * j + 1))) ? +1 : -1;
      im *= (i & (1 << (2 * j))) ? +1 : -1;
    }
    t[86 + i] = make_float2(re / sqrtf(170), im / sqrtf(170));
  }
}

} // namespace

namespace phyhip {
namespace modem {
bool params_for(Params& p, int llr_type)
{
  return fill_params(p, llr_type);
}
const float2* mod_tables()
{
  std::lock_guard<std::mutex> lk(g_mod.mu);
  if (!g_mod.d && !g_mod.failed) {
    std::vector<float2> t;
    build_mod_tables(t);
    if (hipMalloc(&g_mod.d, t.size() * sizeof(float2)) != hipSuccess || upload(g_mod.d, t.data(), t.size() * sizeof(float2)) != hipSuccess) {
      set_error("modem: cannot allocate the constellation tables on the device");
      g_mod.failed = true;
      g_mod.d      = nullptr;
    }
  }
  return g_mod.d;
}
void host_mod_table(uint32_t mod, float2* out)
{
  std::vector<float2> t;
  build_mod_tables(t);
  memcpy(out, t.data() + mod_table_offset(mod), sizeof(float2) << (mod == 0 ? 1 : 2 * mod));
}
} // namespace modem
} // namespace phyhip

namespace {

// ---- host-pointer calls: one job, thread-local staging ----------------------------------------------------------------------
struct Stage {
  hipStream_t st       = nullptr;
  void*       d_in     = nullptr;
  void*       d_out    = nullptr;
  size_t      cap_in   = 0;
  size_t      cap_out  = 0;
  bool        tried    = false;
  ~Stage()
  {
    (void)hipHostFree(d_in);
    (void)hipHostFree(d_out);
    if (st) {
      (void)hipStreamDestroy(st);
    }
  }
  bool ready()
  {
    if (!tried) {
      tried = true;
      if (device_available() && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        st = nullptr;
      }
    }
    return st != nullptr;
  }
  static bool grow(void** p, size_t* cap, size_t need)
  {
    if (need <= *cap) {
      return true;
    }
    // PINNED HOST memory, mapped into the device's address space: a single call's symbols are read once and its soft bits written once, so the
    // kernel works on the staging images themselves -- no copy operation on either side of it (6-9 us each whatever the size, and from / to
    // pageable memory a staged, blocking one: tools/probe/roundtrip_probe.hip)
    (void)hipHostFree(*p);
    *p   = nullptr;
    *cap = 0;
    if (host_image_alloc(p, need + need / 2 + 256) != hipSuccess) {
      return false;
    }
    *cap = need + need / 2 + 256;
    return true;
  }
};

Stage& stage()
{
  return thread_device_local<Stage>(); // this thread's staging context on the device it is bound to
}

int run_host(uint32_t mod, const void* in, void* out, int llr_type, uint32_t n, uint32_t seed, bool scramble, const char* who)
{
  if (n == 0) {
    return SRSRAN_SUCCESS;
  }
  Stage& s = stage();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] %s: no HIP device (there is no CPU fallback)\n", who);
    return SRSRAN_ERROR;
  }
  const size_t   es       = llr_size(llr_type);
  const size_t   in_bytes = mod == modem::MOD_PASS ? n * es : (size_t)n * sizeof(cf_t);
  const size_t   n_llr    = (size_t)n * bits_per_symbol(mod);
  if (scramble && n_llr > SRSRAN_HIP_SEQUENCE_MAX_LEN) {
    fprintf(stderr, "[srsran_phy_hip] %s: %zu soft bits exceed the sequence tables (%u)\n", who, n_llr, SRSRAN_HIP_SEQUENCE_MAX_LEN);
    return SRSRAN_ERROR;
  }
  modem::Params p;
  if (!fill_params(p, llr_type) || !Stage::grow(&s.d_in, &s.cap_in, in_bytes) || !Stage::grow(&s.d_out, &s.cap_out, n_llr * es)) {
    fprintf(stderr, "[srsran_phy_hip] %s: %s\n", who, get_error());
    return SRSRAN_ERROR;
  }
  p.in      = s.d_in;
  p.out     = s.d_out;
  p.single  = modem::Job{mod, n, 0, 0, seed, scramble ? 1u : 0u, 0, modem::tiles_of(mod, n)};
  p.n_jobs  = 1;
  p.n_tiles = p.single.ntiles;
  memcpy(s.d_in, in, in_bytes);
  PHY_HIP_CHECK(modem::launch(p, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  memcpy(out, s.d_out, n_llr * es);
  return SRSRAN_SUCCESS;
}

int demod_host(srsran_mod_t m, const cf_t* symbols, void* llr, int llr_type, int nsymbols, const char* who)
{
  if ((int)m < 0 || (int)m > SRSRAN_MOD_256QAM) {
    fprintf(stderr, "Invalid modulation %d\n", (int)m); // demod_soft.c:865,890,915
    return -1;
  }
  if (nsymbols <= 0) {
    return 0;
  }
  return run_host((uint32_t)m, symbols, llr, llr_type, (uint32_t)nsymbols, 0, false, who) == SRSRAN_SUCCESS ? 0 : -1;
}

inline uint32_t pdsch_seed(uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id)
{
  return ((uint32_t)rnti << 14) + ((uint32_t)q << 13) + ((nslot / 2) << 9) + cell_id;
}
inline uint32_t pusch_seed(uint16_t rnti, uint32_t nslot, uint32_t cell_id)
{
  return ((uint32_t)rnti << 14) + ((nslot / 2) << 9) + cell_id;
}

} // namespace

extern "C" int srsran_demod_soft_demodulate(srsran_mod_t m, const cf_t* symbols, float* llr, int nsymbols)
{
  return demod_host(m, symbols, llr, modem::LLR_F32, nsymbols, "srsran_demod_soft_demodulate");
}
extern "C" int srsran_demod_soft_demodulate_s(srsran_mod_t m, const cf_t* symbols, short* llr, int nsymbols)
{
  return demod_host(m, symbols, llr, modem::LLR_I16, nsymbols, "srsran_demod_soft_demodulate_s");
}
extern "C" int srsran_demod_soft_demodulate_b(srsran_mod_t m, const cf_t* symbols, int8_t* llr, int nsymbols)
{
  return demod_host(m, symbols, llr, modem::LLR_I8, nsymbols, "srsran_demod_soft_demodulate_b");
}

extern "C" void srsran_sequence_apply_f(const float* in, float* out, uint32_t length, uint32_t seed)
{
  (void)run_host(modem::MOD_PASS, in, out, modem::LLR_F32, length, seed, true, "srsran_sequence_apply_f");
}
extern "C" void srsran_sequence_apply_s(const int16_t* in, int16_t* out, uint32_t length, uint32_t seed)
{
  (void)run_host(modem::MOD_PASS, in, out, modem::LLR_I16, length, seed, true, "srsran_sequence_apply_s");
}
extern "C" void srsran_sequence_apply_c(const int8_t* in, int8_t* out, uint32_t length, uint32_t seed)
{
  (void)run_host(modem::MOD_PASS, in, out, modem::LLR_I8, length, seed, true, "srsran_sequence_apply_c");
}
extern "C" void srsran_sequence_pdsch_apply_f(const float* in, float* out, uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id, uint32_t len)
{
  srsran_sequence_apply_f(in, out, len, pdsch_seed(rnti, q, nslot, cell_id));
}
extern "C" void srsran_sequence_pdsch_apply_s(const int16_t* in, int16_t* out, uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id, uint32_t len)
{
  srsran_sequence_apply_s(in, out, len, pdsch_seed(rnti, q, nslot, cell_id));
}
extern "C" void srsran_sequence_pdsch_apply_c(const int8_t* in, int8_t* out, uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id, uint32_t len)
{
  srsran_sequence_apply_c(in, out, len, pdsch_seed(rnti, q, nslot, cell_id));
}
extern "C" void srsran_sequence_pusch_apply_s(const int16_t* in, int16_t* out, uint16_t rnti, uint32_t nslot, uint32_t cell_id, uint32_t len)
{
  srsran_sequence_apply_s(in, out, len, pusch_seed(rnti, nslot, cell_id));
}
extern "C" void srsran_sequence_pusch_apply_c(const int8_t* in, int8_t* out, uint16_t rnti, uint32_t nslot, uint32_t cell_id, uint32_t len)
{
  srsran_sequence_apply_c(in, out, len, pusch_seed(rnti, nslot, cell_id));
}
extern "C" uint32_t srsran_hip_sequence_pdsch_seed(uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id)
{
  return pdsch_seed(rnti, q, nslot, cell_id);
}
extern "C" uint32_t srsran_hip_sequence_pusch_seed(uint16_t rnti, uint32_t nslot, uint32_t cell_id)
{
  return pusch_seed(rnti, nslot, cell_id);
}

// ---- batched, device resident --------------------------------------------------------------------------------------------------
struct srsran_hip_demod {
  DeviceTag tag;
  modem::Job* d_jobs  = nullptr;
  modem::Job* h_jobs  = nullptr; // pinned
  size_t      cap     = 0;
  uint32_t*   d_map   = nullptr; // job of every tile
  uint32_t*   h_map   = nullptr; // pinned
  size_t      map_cap = 0;
  hipEvent_t  done    = nullptr; // the previous call's kernel has consumed d_jobs / h_jobs
  bool        pending = false;
};

extern "C" int srsran_hip_demod_create(srsran_hip_demod_t** hh)
{
  if (!hh) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  srsran_hip_demod* h = new srsran_hip_demod;
  if (hipEventCreateWithFlags(&h->done, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return SRSRAN_ERROR;
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_demod_free(srsran_hip_demod_t* h)
{
  if (!h) {
    return;
  }
  if (h->pending) {
    (void)hipEventSynchronize(h->done);
  }
  (void)hipFree(h->d_jobs);
  (void)hipHostFree(h->h_jobs);
  (void)hipFree(h->d_map);
  (void)hipHostFree(h->h_map);
  (void)hipEventDestroy(h->done);
  delete h;
}

extern "C" int srsran_hip_demod_run(srsran_hip_demod_t* h, const void* d_in, void* d_llr, int llr_type,
                                    const srsran_hip_demod_job_t* jobs, uint32_t n_jobs, void* stream)
{
  TraceRange trace_("srsran_hip_demod_run");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_demod_run", SRSRAN_ERROR);
  }
  if (!h || (n_jobs && (!jobs || !d_in || !d_llr)) || llr_type < 0 || llr_type > 2) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (n_jobs == 0) {
    return SRSRAN_SUCCESS;
  }
  for (uint32_t i = 0; i < n_jobs; i++) {
    const srsran_hip_demod_job_t& j = jobs[i];
    if (j.mod > SRSRAN_HIP_MOD_NONE || ((j.descramble & 1u) && (uint64_t)j.nof_symbols * bits_per_symbol(j.mod) > SRSRAN_HIP_SEQUENCE_MAX_LEN)) {
      set_error("srsran_hip_demod_run: job %u: modulation %u / %u symbols not supported", i, j.mod, j.nof_symbols);
      fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (h->pending) {
    PHY_HIP_CHECK(hipEventSynchronize(h->done), SRSRAN_ERROR);
    h->pending = false;
  }
  if (n_jobs > h->cap) {
    (void)hipFree(h->d_jobs);
    (void)hipHostFree(h->h_jobs);
    h->d_jobs = nullptr;
    h->h_jobs = nullptr;
    h->cap    = 0;
    const size_t cap = (size_t)n_jobs + n_jobs / 2 + 16;
    PHY_HIP_CHECK(hipMalloc(&h->d_jobs, cap * sizeof(modem::Job)), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipHostMalloc(&h->h_jobs, cap * sizeof(modem::Job)), SRSRAN_ERROR);
    h->cap = cap;
  }
  modem::Params p;
  if (!fill_params(p, llr_type)) {
    return SRSRAN_ERROR;
  }
  uint32_t tiles = 0;
  for (uint32_t i = 0; i < n_jobs; i++) {
    const srsran_hip_demod_job_t& j = jobs[i];
    const uint32_t                nt = modem::tiles_of(j.mod, j.nof_symbols);
    h->h_jobs[i] = modem::Job{j.mod, j.nof_symbols, j.symbol_offset, j.llr_offset, j.seed, j.descramble & 3u, tiles, nt};
    tiles += nt;
  }
  if (tiles > h->map_cap) {
    (void)hipFree(h->d_map);
    (void)hipHostFree(h->h_map);
    h->d_map   = nullptr;
    h->h_map   = nullptr;
    h->map_cap = 0;
    const size_t cap = (size_t)tiles + tiles / 2 + 256;
    PHY_HIP_CHECK(hipMalloc(&h->d_map, cap * sizeof(uint32_t)), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipHostMalloc(&h->h_map, cap * sizeof(uint32_t)), SRSRAN_ERROR);
    h->map_cap = cap;
  }
  for (uint32_t i = 0, t = 0; i < n_jobs; i++) {
    for (uint32_t k = 0; k < h->h_jobs[i].ntiles; k++) {
      h->h_map[t++] = i;
    }
  }
  PHY_HIP_CHECK(hipMemcpyAsync(h->d_map, h->h_map, tiles * sizeof(uint32_t), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  p.tile_job = h->d_map;
  p.in      = d_in;
  p.out     = d_llr;
  p.jobs    = h->d_jobs;
  p.n_jobs  = n_jobs;
  p.n_tiles = tiles;
  PHY_HIP_CHECK(hipMemcpyAsync(h->d_jobs, h->h_jobs, n_jobs * sizeof(modem::Job), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  PHY_HIP_CHECK(modem::launch(p, st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipEventRecord(h->done, st), SRSRAN_ERROR);
  h->pending = true;
  return SRSRAN_SUCCESS;
}

// ---- single-antenna equaliser ---------------------------------------------------------------------------------------------------
extern "C" int srsran_hip_predecoding_single(const cf_t* d_y, const cf_t* d_h, cf_t* d_x, float* d_csi, uint32_t nof_symbols, float scaling,
                                             float noise_estimate, void* stream)
{
  if (nof_symbols && (!d_y || !d_h || !d_x || ((((uintptr_t)d_y) | ((uintptr_t)d_h) | ((uintptr_t)d_x)) & 15u) || (((uintptr_t)d_csi) & 7u))) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(modem::launch_eq(d_y, d_h, d_x, d_csi, nof_symbols, scaling, noise_estimate, (hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_predecoding_single(cf_t* y, cf_t* h, cf_t* x, float* csi, int nof_symbols, float scaling, float noise_estimate)
{
  if (nof_symbols <= 0) {
    return nof_symbols;
  }
  Stage& s = stage();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_predecoding_single: no HIP device (there is no CPU fallback)\n");
    return SRSRAN_ERROR;
  }
  const size_t nb   = (size_t)nof_symbols * sizeof(cf_t);
  const size_t slot = (nb + 255) & ~(size_t)255;
  if (!Stage::grow(&s.d_in, &s.cap_in, 2 * slot) || !Stage::grow(&s.d_out, &s.cap_out, slot + (size_t)nof_symbols * sizeof(float))) {
    return SRSRAN_ERROR;
  }
  uint8_t* din  = (uint8_t*)s.d_in;
  uint8_t* dout = (uint8_t*)s.d_out;
  memcpy(din, y, nb); // (the staging images are pinned host memory the kernel works on directly, see Stage::grow)
  memcpy(din + slot, h, nb);
  PHY_HIP_CHECK(modem::launch_eq(din, din + slot, dout, csi ? (float*)(dout + slot) : nullptr, (uint32_t)nof_symbols, scaling, noise_estimate, s.st),
                SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  memcpy(x, dout, nb);
  if (csi) {
    memcpy(csi, dout + slot, (size_t)nof_symbols * sizeof(float));
  }
  return nof_symbols;
}
