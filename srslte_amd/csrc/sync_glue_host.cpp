// sync_glue_host.cpp -- srsran_cexptab_*, srsran_cfo_*, srsran_cp_synch_* and srsran_sync_* of the drop-in ABI.
//
// Mirrors (interface + behaviour) lib/src/phy/utils/cexptab.c, lib/src/phy/sync/cfo.c, cp.c and sync.c of the
// reference.  srsran_sync_find is a chain of short stages over ONE frame (integer CFO trial, CP-based CFO,
// PSS search, PSS-based CFO, SSS, CP detection); each stage runs on the device through the PSS / SSS / DFT
// handles and the small kernels of sync_kernels.hip.  The throughput path for many captures is
// srsran_hip_cellsearch_* (sync_host.cpp); this file is the per-frame control flow the reference's callers
// (ue_sync.c, ue_cell_search.c) expect.
#include "hip_common.h"
#include "srsran_amd/phy_sync_abi.h"
#include "sync_device.h"
#include "sync_glue.h"

#include <cmath>
#include <complex>

using namespace phyhip;

// ------------------------------------------------------------------------------------------------ device staging

namespace {

struct Stage {
  hipStream_t st   = nullptr;
  float2*     d[3] = {nullptr, nullptr, nullptr};
  size_t      cap[3] = {0, 0, 0};
  float2*     d_small = nullptr; // 8 results + arg-max
  ~Stage()
  {
    for (auto* p : d) {
      (void)hipFree(p);
    }
    (void)hipFree(d_small);
    if (st) {
      (void)hipStreamDestroy(st);
    }
  }
  bool ready()
  {
    if (st) {
      return true;
    }
    if (!device_available()) {
      return false;
    }
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&d_small, 16 * sizeof(float2)) != hipSuccess) {
      set_error("sync glue: cannot create the staging stream");
      st = nullptr;
      return false;
    }
    return true;
  }
  float2* buf(int i, size_t n)
  {
    if (n > cap[i]) {
      (void)hipFree(d[i]);
      d[i]   = nullptr;
      cap[i] = 0;
      if (hipMalloc(&d[i], n * sizeof(float2)) != hipSuccess) {
        set_error("sync glue: device allocation of %zu samples failed", n);
        return nullptr;
      }
      cap[i] = n;
    }
    return d[i];
  }
};

Stage& stage()
{
  Stage& s = thread_device_local<Stage>();
  return s;
}

} // namespace

int glue::prod(const cf_t* a, const cf_t* b, cf_t* out, int n, bool conj_b)
{
  Stage& s = stage();
  if (n <= 0) {
    return SRSRAN_SUCCESS;
  }
  if (!s.ready()) {
    return SRSRAN_ERROR;
  }
  float2 *da = s.buf(0, n), *db = s.buf(1, n), *dc = s.buf(2, n);
  if (!da || !db || !dc) {
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(da, a, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(db, b, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(sync::launch_cmul(da, db, dc, n, conj_b, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(out, dc, n * sizeof(cf_t), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

int glue::lincomb(const cf_t* a, float sa, const cf_t* b, float sb, cf_t* out, int n)
{
  Stage& s = stage();
  if (n <= 0) {
    return SRSRAN_SUCCESS;
  }
  if (!s.ready()) {
    return SRSRAN_ERROR;
  }
  float2 *da = s.buf(0, n), *db = s.buf(1, n), *dc = s.buf(2, n);
  if (!da || !db || !dc) {
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(da, a, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(db, b, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(sync::launch_lincomb(da, sa, db, sb, dc, n, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(out, dc, n * sizeof(cf_t), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

int glue::apply_cfo(const cf_t* in, cf_t* out, int n, float cfo)
{
  Stage& s = stage();
  if (n <= 0) {
    return SRSRAN_SUCCESS;
  }
  if (!s.ready()) {
    return SRSRAN_ERROR;
  }
  float2 *da = s.buf(0, n), *dc = s.buf(2, n);
  if (!da || !dc) {
    return SRSRAN_ERROR;
  }
  // oscillator seeds exactly as vector_simd.c:1694-1706,1732-1733 forms them (float products, then cexpf)
  const float    TWOPI = 2.0f * (float)M_PI;
  sync::CfoSeeds sd;
  for (int k = 0; k < 8; k++) {
    const float t  = TWOPI * cfo * (float)k;
    sd.phase_re[k] = cosf(t);
    sd.phase_im[k] = sinf(t);
  }
  const float t8 = TWOPI * cfo * 8.0f, t1 = TWOPI * cfo, tt = TWOPI * cfo * (float)(n >= 8 ? (n / 8) * 8 : 0);
  sd.osc8_re = cosf(t8);
  sd.osc8_im = sinf(t8);
  sd.osc1_re = cosf(t1);
  sd.osc1_im = sinf(t1);
  sd.tail_re = cosf(tt);
  sd.tail_im = sinf(tt);
  PHY_HIP_CHECK(hipMemcpyAsync(da, in, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(sync::launch_apply_cfo(da, dc, n, sd, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(out, dc, n * sizeof(cf_t), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

int glue::dots(const Dot* jobs, int count, cf_t* results)
{
  Stage& s = stage();
  if (count <= 0 || count > 8) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!s.ready()) {
    return SRSRAN_ERROR;
  }
  size_t total = 0;
  for (int j = 0; j < count; j++) {
    total += (size_t)jobs[j].n * (jobs[j].mode == POWER ? 1 : 2);
  }
  float2* base = s.buf(0, total ? total : 1);
  if (!base) {
    return SRSRAN_ERROR;
  }
  sync::DotJobs dj = {};
  dj.count         = count;
  size_t at        = 0;
  for (int j = 0; j < count; j++) {
    const size_t n = (size_t)jobs[j].n;
    dj.n[j]        = jobs[j].n;
    dj.mode[j]     = jobs[j].mode;
    dj.a[j]        = base + at;
    if (n) {
      PHY_HIP_CHECK(hipMemcpyAsync(base + at, jobs[j].a, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
    }
    at += n;
    dj.b[j] = base + at;
    if (jobs[j].mode != POWER) {
      if (n) {
        PHY_HIP_CHECK(hipMemcpyAsync(base + at, jobs[j].b, n * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
      }
      at += n;
    }
  }
  PHY_HIP_CHECK(sync::launch_dots(dj, s.d_small, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(results, s.d_small, count * sizeof(cf_t), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

int glue::cp_synch(const cf_t* in, int n_in, cf_t* corr, int max_offset, int nof_symbols, int cp_len, int N, uint32_t* argmax)
{
  Stage& s = stage();
  *argmax  = 0;
  if (max_offset <= 0) {
    return SRSRAN_SUCCESS;
  }
  if (!s.ready()) {
    return SRSRAN_ERROR;
  }
  float2 *din = s.buf(0, n_in), *dcorr = s.buf(1, max_offset);
  if (!din || !dcorr) {
    return SRSRAN_ERROR;
  }
  int* d_arg = reinterpret_cast<int*>(s.d_small + 8);
  int  h_arg = 0;
  PHY_HIP_CHECK(hipMemcpyAsync(din, in, (size_t)n_in * sizeof(cf_t), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(sync::launch_cp_synch(din, dcorr, max_offset, nof_symbols, cp_len, N, d_arg, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(corr, dcorr, (size_t)max_offset * sizeof(cf_t), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(&h_arg, d_arg, sizeof(int), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  *argmax = (uint32_t)h_arg;
  return SRSRAN_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ cexptab (tables: host)

extern "C" int srsran_cexptab_init(srsran_cexptab_t* h, uint32_t size)
{
  h->size = size;
  h->tab  = (cf_t*)malloc(sizeof(cf_t) * (1 + (size_t)size));
  if (!h->tab) {
    return SRSRAN_ERROR;
  }
  for (uint32_t i = 0; i < size; i++) {
    // cexptab.c:40: phase in double (M_PI), rounded to float before cexpf
    const float ph = (float)(2 * M_PI * (double)(float)i / (double)size);
    h->tab[i]      = cf_t(cosf(ph), sinf(ph));
  }
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_cexptab_free(srsran_cexptab_t* h)
{
  free(h->tab);
  memset(h, 0, sizeof(*h));
}

extern "C" void srsran_cexptab_gen(srsran_cexptab_t* h, cf_t* x, float freq, uint32_t len)
{
  // cexptab.c:56-74: float phase accumulator, truncated to a table index.  A table build (like twiddles), not a
  // per-sample signal path: the accumulation is sequential by definition.
  const float size      = (float)h->size;
  const float phase_inc = freq * size;
  float       phase     = 0;
  for (uint32_t i = 0; i < len; i++) {
    while (phase >= size) {
      phase -= size;
    }
    while (phase < 0) {
      phase += size;
    }
    x[i] = h->tab[(uint32_t)phase];
    phase += phase_inc;
  }
}

extern "C" void srsran_cexptab_gen_direct(cf_t* x, float freq, uint32_t len)
{
  for (uint32_t i = 0; i < len; i++) {
    const float ph = (float)(2 * M_PI * (double)freq * (double)i);
    x[i]           = cf_t(cosf(ph), sinf(ph));
  }
}

extern "C" void srsran_cexptab_gen_sf(cf_t* x, float freq, uint32_t fft_size)
{
  cf_t* ptr = x;
  for (uint32_t n = 0; n < 2; n++) {
    for (uint32_t i = 0; i < 7; i++) {
      const uint32_t cplen = (uint32_t)ceilf(((float)(i == 0 ? 160 : 144) * fft_size) / 2048.0f);
      for (uint32_t t = 0; t < fft_size + cplen; t++) {
        const float ph = (float)(2 * M_PI * (double)((float)t - (float)cplen) * (double)freq / (double)fft_size);
        ptr[t]         = cf_t(cosf(ph), sinf(ph));
      }
      ptr += fft_size + cplen;
    }
  }
}

// ------------------------------------------------------------------------------------------------ srsran_cfo_t

// The reference is built with SRSRAN_CFO_USE_EXP_TABLE 0 (cfo.c:33): srsran_cfo_correct applies the CFO with
// srsran_vec_apply_cfo over the sample count given at INIT (srsran_cfo_resize and the tolerance are no-ops in that
// build).  The table is still kept for srsran_cfo_correct_offset, which uses it unconditionally (cfo.c:120-134).
extern "C" int srsran_cfo_init(srsran_cfo_t* h, uint32_t nsamples)
{
  memset(h, 0, sizeof(*h));
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  if (srsran_cexptab_init(&h->tab, SRSRAN_CFO_CEXPTAB_SIZE)) {
    return SRSRAN_ERROR;
  }
  h->cur_cexp = (cf_t*)malloc(sizeof(cf_t) * (nsamples ? nsamples : 1));
  if (!h->cur_cexp) {
    srsran_cfo_free(h);
    return SRSRAN_ERROR;
  }
  h->nsamples    = (int)nsamples;
  h->max_samples = (int)nsamples;
  srsran_cexptab_gen(&h->tab, h->cur_cexp, h->last_freq, (uint32_t)h->nsamples);
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_cfo_free(srsran_cfo_t* h)
{
  srsran_cexptab_free(&h->tab);
  free(h->cur_cexp);
  memset(h, 0, sizeof(*h));
}

extern "C" void srsran_cfo_set_tol(srsran_cfo_t* h, float tol)
{
  h->tol = tol;
}

extern "C" int srsran_cfo_resize(srsran_cfo_t* h, uint32_t samples)
{
  (void)h;
  (void)samples;
  return SRSRAN_SUCCESS; // cfo.c:83-95 with SRSRAN_CFO_USE_EXP_TABLE 0: nothing changes, not even nsamples
}

extern "C" void srsran_cfo_correct(srsran_cfo_t* h, const cf_t* input, cf_t* output, float freq)
{
  if (glue::apply_cfo(input, output, h->nsamples, freq)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_cfo_correct: %s\n", get_error());
  }
}

extern "C" void srsran_cfo_correct_offset(srsran_cfo_t* h, const cf_t* input, cf_t* output, float freq, int cexp_offset, int nsamples)
{
  if (fabs(h->last_freq - freq) > h->tol) { // cfo.c:126-130
    h->last_freq = freq;
    srsran_cexptab_gen(&h->tab, h->cur_cexp, h->last_freq, (uint32_t)h->nsamples);
  }
  if (glue::prod(&h->cur_cexp[cexp_offset], input, output, nsamples, false)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_cfo_correct_offset: %s\n", get_error());
  }
}

// cfo.c:130-151: CP-based CFO estimate of an uplink subframe (one symbol's cyclic prefix against its tail), in place: the half-sub-carrier shift of
// SC-FDMA is taken out first and put back, corrected by the estimate, at the end.  Three device round trips (the reference's three vector calls).
extern "C" float srsran_cfo_est_corr_cp(cf_t* input_buffer, uint32_t nof_prb)
{
  const int nFFT = srsran_symbol_sz(nof_prb);
  if (!input_buffer || nFFT <= 0) {
    return 0.f;
  }
  const int   sf_n_samples = nFFT * 15;
  const float tFFT         = (float)(1 / 15000.0);
  auto        cp_norm      = [&](int symbol) { return (symbol == 0 ? 160 * nFFT + 2047 : 144 * nFFT + 2047) / 2048; }; // SRSRAN_CP_LEN_NORM, phy_common.h:125-128
  const int   cp_size      = cp_norm(1);
  if (glue::apply_cfo(input_buffer, input_buffer, sf_n_samples, (float)(1 / (nFFT * 15e3)) * (float)(15e3 / 2.0))) {
    fprintf(stderr, "[srsran_phy_hip] srsran_cfo_est_corr_cp: %s\n", get_error());
    return 0.f;
  }
  glue::Dot job = {&input_buffer[nFFT + cp_norm(0)], &input_buffer[2 * nFFT + cp_norm(0)], cp_size, glue::CONJ};
  cf_t      est(0.f, 0.f);
  if (glue::dots(&job, 1, &est)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_cfo_est_corr_cp: %s\n", get_error());
    return 0.f;
  }
  const float cfo = (float)(-1 * atan2f(est.imag(), est.real()) / (float)(2 * M_PI * tFFT));
  if (glue::apply_cfo(input_buffer, input_buffer, sf_n_samples, (float)(1 / (nFFT * 15e3)) * (float)((-15e3 / 2.0) - cfo))) {
    fprintf(stderr, "[srsran_phy_hip] srsran_cfo_est_corr_cp: %s\n", get_error());
  }
  return cfo;
}

// ------------------------------------------------------------------------------------------------ srsran_cp_synch_t

extern "C" int srsran_cp_synch_init(srsran_cp_synch_t* q, uint32_t symbol_sz)
{
  if (!q) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  memset(q, 0, sizeof(*q));
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  q->corr = (cf_t*)calloc(symbol_sz ? symbol_sz : 1, sizeof(cf_t));
  if (!q->corr) {
    perror("malloc");
    return SRSRAN_ERROR;
  }
  q->symbol_sz     = symbol_sz;
  q->max_symbol_sz = symbol_sz;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_cp_synch_free(srsran_cp_synch_t* q)
{
  if (q) {
    free(q->corr);
    q->corr = nullptr;
  }
}

extern "C" int srsran_cp_synch_resize(srsran_cp_synch_t* q, uint32_t symbol_sz)
{
  if (!q) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (symbol_sz > q->max_symbol_sz) {
    fprintf(stderr, "Error in cp_synch_resize(): symbol_sz must be lower than initialized\n");
    return SRSRAN_ERROR;
  }
  q->symbol_sz = symbol_sz;
  return SRSRAN_SUCCESS;
}

extern "C" uint32_t srsran_cp_synch(srsran_cp_synch_t* q, const cf_t* input, uint32_t max_offset, uint32_t nof_symbols, uint32_t cp_len)
{
  if (max_offset > q->symbol_sz) {
    max_offset = q->symbol_sz;
  }
  // samples touched: offset max_offset-1, nof_symbols symbols of N + cp (+1 on every 7th), plus N for the partner
  const uint32_t n_in = max_offset + nof_symbols * (q->symbol_sz + cp_len + 1) + q->symbol_sz;
  uint32_t       idx  = 0;
  if (glue::cp_synch(input, (int)n_in, q->corr, (int)max_offset, (int)nof_symbols, (int)cp_len, (int)q->symbol_sz, &idx)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_cp_synch: %s\n", get_error());
  }
  return idx;
}

extern "C" cf_t srsran_cp_synch_corr_output(srsran_cp_synch_t* q, uint32_t offset)
{
  return offset < q->symbol_sz ? q->corr[offset] : cf_t(0, 0);
}

// ------------------------------------------------------------------------------------------------ srsran_sync_t

namespace {

const float kCfoEmaAlpha = 0.1f, kCpEmaAlpha = 0.1f; // sync.c:34-35
const float kMaxCfoPssOffsetHz = 7000.f;              // sync.c:39

int cp_samples(uint32_t N, int c)
{
  return (int)ceilf((((float)(c) * (N)) / 2048.0f)); // SRSRAN_CP_LEN, phy_common.h:125
}

bool fft_size_ok(uint32_t n)
{
  return n >= SRSRAN_SYNC_FFT_SZ_MIN && n <= SRSRAN_SYNC_FFT_SZ_MAX && (n % 64) == 0;
}

float ema(float data, float average, float alpha)
{
  return alpha * data + (1 - alpha) * average; // SRSRAN_VEC_EMA, vector.h:63
}

void fill_cfo_i_tables(srsran_sync_t* q)
{
  for (int i = 0; i < 2; i++) {
    const int offset = i == 0 ? -1 : 1;
    for (uint32_t t = 0; t < q->frame_size; t++) {
      // sync.c:224,373: cexpf(-2 j pi offset t / fft_size), argument evaluated in double then rounded
      const float ph      = (float)(-2.0 * M_PI * (double)offset * (double)(float)t / (double)q->fft_size);
      q->cfo_i_corr[i][t] = cf_t(cosf(ph), sinf(ph));
    }
  }
}

// frequency-domain SSS of a known cell as the receiver's filtered symbol sees it (sync.c:311-336)
void make_known_sss(srsran_sync_t* q, uint32_t N_id_1)
{
  float sf[2][SRSRAN_SSS_LEN];
  cf_t  symbol[SRSRAN_SYMBOL_SZ_MAX];
  q->N_id_1 = N_id_1;
  srsran_sss_generate(sf[0], sf[1], q->N_id_1 * 3 + q->N_id_2);
  const uint32_t k = q->fft_size / 2 - 31;
  for (int n = 0; n < 2; n++) {
    for (uint32_t i = 0; i < q->fft_size; i++) {
      symbol[i] = cf_t(0, 0);
    }
    for (uint32_t i = 0; i < SRSRAN_SSS_LEN; i++) {
      symbol[k + i] = cf_t(sf[n][i], 0);
    }
    srsran_dft_run_c(&q->idftp_sss, symbol, q->sss_signal[n]);
  }
  q->sss_generated = true;
}

} // namespace

extern "C" int srsran_sync_init(srsran_sync_t* q, uint32_t frame_size, uint32_t max_offset, uint32_t fft_size)
{
  return srsran_sync_init_decim(q, frame_size, max_offset, fft_size, 1);
}

extern "C" int srsran_sync_init_decim(srsran_sync_t* q, uint32_t frame_size, uint32_t max_offset, uint32_t fft_size, int decimate)
{
  if (q == NULL || !fft_size_ok(fft_size)) {
    fprintf(stderr, "Invalid parameters frame_size: %d, fft_size: %d\n", frame_size, fft_size);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  memset(q, 0, sizeof(srsran_sync_t));
  q->N_id_2            = 1000;
  q->N_id_1            = 1000;
  q->cfo_ema_alpha     = kCfoEmaAlpha;
  q->sss_alg           = SSS_FULL;
  q->detect_cp         = true;
  q->sss_en            = true;
  q->detect_frame_type = true;
  q->cfo_cp_nsymbols   = 3;
  q->fft_size          = fft_size;
  q->frame_size        = frame_size;
  q->max_offset        = max_offset;
  q->max_frame_size    = frame_size;
  srsran_sync_cfo_reset(q, 0.0f);

  bool ok = srsran_cfo_init(&q->cfo_corr_frame, q->frame_size) == SRSRAN_SUCCESS &&
            srsran_cfo_init(&q->cfo_corr_symbol, q->fft_size) == SRSRAN_SUCCESS;
  if (ok) {
    srsran_sync_set_cfo_tol(q, 0.0f); // DEFAULT_CFO_TOL, sync.c:37
    for (int i = 0; i < 2 && ok; i++) {
      q->cfo_i_corr[i] = (cf_t*)calloc(q->frame_size ? q->frame_size : 1, sizeof(cf_t));
      ok               = q->cfo_i_corr[i] != nullptr;
    }
    q->temp = (cf_t*)calloc(2 * (size_t)q->frame_size + 1, sizeof(cf_t));
    ok      = ok && q->temp != nullptr;
  }
  if (ok) {
    srsran_sync_set_cp(q, SRSRAN_CP_NORM);
    q->decimate = decimate;
    if (!decimate) {
      decimate = 1;
    }
    ok = srsran_dft_plan(&q->idftp_sss, (int)fft_size, SRSRAN_DFT_BACKWARD, SRSRAN_DFT_COMPLEX) == SRSRAN_SUCCESS;
  }
  if (ok) {
    srsran_dft_plan_set_mirror(&q->idftp_sss, true);
    srsran_dft_plan_set_dc(&q->idftp_sss, true);
    srsran_dft_plan_set_norm(&q->idftp_sss, false);
    ok = srsran_pss_init_fft_offset_decim(&q->pss, max_offset, fft_size, 0, decimate) == SRSRAN_SUCCESS &&
         srsran_sss_init(&q->sss, fft_size) == SRSRAN_SUCCESS && srsran_cp_synch_init(&q->cp_synch, fft_size) == SRSRAN_SUCCESS;
  }
  if (!ok) {
    fprintf(stderr, "[srsran_phy_hip] srsran_sync_init: %s\n", get_error());
    srsran_sync_free(q);
    return SRSRAN_ERROR;
  }
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_sync_free(srsran_sync_t* q)
{
  if (!q) {
    return;
  }
  if (q->pss.conv_fft.input_fft) {
    srsran_pss_free(&q->pss);
  }
  if (q->sss.dftp_input.p) {
    srsran_sss_free(&q->sss);
  }
  srsran_cfo_free(&q->cfo_corr_frame);
  srsran_cfo_free(&q->cfo_corr_symbol);
  srsran_cp_synch_free(&q->cp_synch);
  if (q->idftp_sss.p) {
    srsran_dft_plan_free(&q->idftp_sss);
  }
  for (int i = 0; i < 2; i++) {
    free(q->cfo_i_corr[i]);
    q->cfo_i_corr[i] = nullptr;
    if (q->pss_i[i].conv_fft.input_fft) {
      srsran_pss_free(&q->pss_i[i]);
    }
  }
  free(q->temp);
  q->temp = nullptr;
}

extern "C" int srsran_sync_resize(srsran_sync_t* q, uint32_t frame_size, uint32_t max_offset, uint32_t fft_size)
{
  if (q == NULL || !fft_size_ok(fft_size)) {
    fprintf(stderr, "Invalid parameters frame_size: %d, fft_size: %d\n", frame_size, fft_size);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (frame_size > q->max_frame_size) {
    fprintf(stderr, "Error in sync_resize(): frame_size must be lower than initialized\n");
    return SRSRAN_ERROR;
  }
  q->fft_size   = fft_size;
  q->frame_size = frame_size;
  q->max_offset = max_offset;
  if (srsran_pss_resize(&q->pss, q->max_offset, q->fft_size, 0) || srsran_sss_resize(&q->sss, q->fft_size) ||
      srsran_dft_replan(&q->idftp_sss, (int)fft_size) || srsran_cp_synch_resize(&q->cp_synch, q->fft_size) ||
      srsran_cfo_resize(&q->cfo_corr_frame, q->frame_size) || srsran_cfo_resize(&q->cfo_corr_symbol, q->fft_size)) {
    fprintf(stderr, "Error resizing the sync object\n");
    return SRSRAN_ERROR;
  }
  if (q->cfo_i_initiated) {
    for (int i = 0; i < 2; i++) {
      if (srsran_pss_resize(&q->pss_i[i], q->max_offset, q->fft_size, i == 0 ? -1 : 1)) {
        fprintf(stderr, "Error initializing PSS object\n");
      }
    }
    fill_cfo_i_tables(q);
  }
  srsran_sync_set_cfo_tol(q, q->current_cfo_tol);
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_sync_reset(srsran_sync_t* q)
{
  q->M_ext_avg  = 0;
  q->M_norm_avg = 0;
  srsran_pss_reset(&q->pss);
}

extern "C" void srsran_sync_set_frame_type(srsran_sync_t* q, srsran_frame_type_t frame_type)
{
  q->frame_type        = frame_type;
  q->detect_frame_type = false;
}

extern "C" void srsran_sync_set_cfo_tol(srsran_sync_t* q, float tol)
{
  q->current_cfo_tol = tol;
  srsran_cfo_set_tol(&q->cfo_corr_frame, (float)(tol / (15000.0 * q->fft_size)));
  srsran_cfo_set_tol(&q->cfo_corr_symbol, (float)(tol / (15000.0 * q->fft_size)));
}

extern "C" void srsran_sync_set_threshold(srsran_sync_t* q, float threshold)
{
  q->threshold = threshold;
}

extern "C" void srsran_sync_sss_en(srsran_sync_t* q, bool enabled)
{
  q->sss_en = enabled;
}

extern "C" bool srsran_sync_sss_detected(srsran_sync_t* q)
{
  return q->sss_detected;
}

extern "C" float srsran_sync_sss_correlation_peak(srsran_sync_t* q)
{
  return q->sss_corr;
}

extern "C" bool srsran_sync_sss_available(srsran_sync_t* q)
{
  return q->sss_available;
}

extern "C" int srsran_sync_get_cell_id(srsran_sync_t* q)
{
  return (q->N_id_2 < 3 && q->N_id_1 < 168) ? (int)(q->N_id_1 * 3 + q->N_id_2) : -1;
}

extern "C" int srsran_sync_set_N_id_2(srsran_sync_t* q, uint32_t N_id_2)
{
  if (N_id_2 >= 3) {
    fprintf(stderr, "Invalid N_id_2=%d\n", N_id_2);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  q->N_id_2 = N_id_2;
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_sync_set_N_id_1(srsran_sync_t* q, uint32_t N_id_1)
{
  if (N_id_1 >= 168) {
    fprintf(stderr, "Invalid N_id_2=%d\n", N_id_1);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  make_known_sss(q, N_id_1);
  return SRSRAN_SUCCESS;
}

extern "C" uint32_t srsran_sync_get_sf_idx(srsran_sync_t* q)
{
  return q->sf_idx;
}

extern "C" float srsran_sync_get_cfo(srsran_sync_t* q)
{
  return q->cfo_cp_mean + q->cfo_pss_mean + q->cfo_i_value;
}

extern "C" void srsran_sync_cfo_reset(srsran_sync_t* q, float init_cfo_hz)
{
  q->cfo_cp_mean    = init_cfo_hz / 15e3f;
  q->cfo_cp_is_set  = false;
  q->cfo_pss_mean   = 0;
  q->cfo_pss_is_set = false;
}

extern "C" void srsran_sync_copy_cfo(srsran_sync_t* q, srsran_sync_t* src_obj)
{
  q->cfo_cp_mean    = src_obj->cfo_cp_mean;
  q->cfo_pss_mean   = src_obj->cfo_pss_mean;
  q->cfo_i_value    = src_obj->cfo_i_value;
  q->cfo_cp_is_set  = false;
  q->cfo_pss_is_set = false;
}

extern "C" void srsran_sync_set_cfo_i_enable(srsran_sync_t* q, bool enable)
{
  q->cfo_i_enable = enable;
  if (q->cfo_i_enable && !q->cfo_i_initiated) {
    for (int i = 0; i < 2; i++) {
      if (srsran_pss_init_fft_offset(&q->pss_i[i], q->max_offset, q->fft_size, i == 0 ? -1 : 1)) {
        fprintf(stderr, "Error initializing PSS object\n");
      }
    }
    fill_cfo_i_tables(q);
    q->cfo_i_initiated = true;
  }
}

extern "C" void srsran_sync_set_sss_eq_enable(srsran_sync_t* q, bool enable)
{
  q->sss_channel_equalize = enable;
  if (enable) {
    q->pss_filtering_enabled = true;
    q->pss.chest_on_filter   = true;
  }
}

extern "C" void srsran_sync_set_pss_filt_enable(srsran_sync_t* q, bool enable)
{
  q->pss_filtering_enabled = enable;
}

extern "C" void srsran_sync_set_cfo_cp_enable(srsran_sync_t* q, bool enable, uint32_t nof_symbols)
{
  q->cfo_cp_enable   = enable;
  q->cfo_cp_nsymbols = nof_symbols;
}

extern "C" void srsran_sync_set_cfo_pss_enable(srsran_sync_t* q, bool enable)
{
  q->cfo_pss_enable = enable;
}

extern "C" void srsran_sync_set_cfo_ema_alpha(srsran_sync_t* q, float alpha)
{
  q->cfo_ema_alpha = alpha;
}

extern "C" float srsran_sync_get_peak_value(srsran_sync_t* q)
{
  return q->peak_value;
}

extern "C" void srsran_sync_cp_en(srsran_sync_t* q, bool enabled)
{
  q->detect_cp = enabled;
}

extern "C" void srsran_sync_set_em_alpha(srsran_sync_t* q, float alpha)
{
  srsran_pss_set_ema_alpha(&q->pss, alpha);
}

extern "C" srsran_cp_t srsran_sync_get_cp(srsran_sync_t* q)
{
  return q->cp;
}

extern "C" void srsran_sync_set_cp(srsran_sync_t* q, srsran_cp_t cp)
{
  q->cp     = cp;
  q->cp_len = (uint32_t)(cp == SRSRAN_CP_NORM ? cp_samples(q->fft_size, 144) : cp_samples(q->fft_size, 512));
  q->nof_symbols = q->frame_size < q->fft_size ? 1 : q->frame_size / (q->fft_size + q->cp_len) - 1;
}

extern "C" void srsran_sync_set_sss_algorithm(srsran_sync_t* q, sss_alg_t alg)
{
  q->sss_alg = alg;
}

extern "C" srsran_pss_t* srsran_sync_get_cur_pss_obj(srsran_sync_t* q)
{
  srsran_pss_t* objs[3] = {&q->pss_i[0], &q->pss, &q->pss_i[1]};
  return objs[q->cfo_i_value + 1];
}

// CP length decision of Kim et al. (sync.c:451-508): CP correlation energy of up to 3 symbols before the peak,
// normal against extended hypothesis, each normalised by the CP power and exponentially averaged over calls.
extern "C" srsran_cp_t srsran_sync_detect_cp(srsran_sync_t* q, const cf_t* input, uint32_t peak_pos)
{
  const uint32_t N = q->fft_size;
  const uint32_t len[2] = {(uint32_t)cp_samples(N, 144), (uint32_t)cp_samples(N, 512)}; // SRSRAN_CP_LEN_NORM(7,.) / _EXT
  uint32_t       nsym   = peak_pos / (N + len[1]);
  nsym                  = nsym > 3 ? 3 : nsym;
  if (nsym == 0) {
    return SRSRAN_CP_NORM;
  }
  float R[2] = {0, 0}, C[2] = {0, 0};
  for (int hyp = 0; hyp < 2; hyp++) {
    glue::Dot jobs[6];
    cf_t      res[6];
    const cf_t* ptr = &input[peak_pos - nsym * (N + len[hyp])];
    for (uint32_t i = 0; i < nsym; i++) {
      jobs[2 * i]     = {&ptr[N], ptr, (int)len[hyp], glue::CONJ};
      jobs[2 * i + 1] = {ptr, nullptr, (int)len[hyp], glue::POWER};
      ptr += N + len[hyp];
    }
    if (glue::dots(jobs, (int)(2 * nsym), res)) {
      fprintf(stderr, "[srsran_phy_hip] srsran_sync_detect_cp: %s\n", get_error());
      return q->cp;
    }
    for (uint32_t i = 0; i < nsym; i++) {
      R[hyp] += res[2 * i].real();
      C[hyp] += (float)len[hyp] * res[2 * i + 1].real();
    }
  }
  const float M_norm = C[0] > 0 ? R[0] / C[0] : 0.f, M_ext = C[1] > 0 ? R[1] / C[1] : 0.f;
  q->M_norm_avg = ema(M_norm / (float)nsym, q->M_norm_avg, kCpEmaAlpha);
  q->M_ext_avg  = ema(M_ext / (float)nsym, q->M_ext_avg, kCpEmaAlpha);
  if (q->M_norm_avg != q->M_ext_avg) {
    return q->M_norm_avg > q->M_ext_avg ? SRSRAN_CP_NORM : SRSRAN_CP_EXT;
  }
  return R[0] > R[1] ? SRSRAN_CP_NORM : SRSRAN_CP_EXT;
}

namespace {

// SSS of one symbol (sync.c:510-574).  With a known N_id_1 only the subframe (0/5) is decided, by comparing
// the correlation with both stored sequences; otherwise full m0/m1 detection.
bool sss_of_symbol(srsran_sync_t* q, const cf_t* input, uint32_t* sf_idx, uint32_t* N_id_1, float* corr)
{
  srsran_sss_set_N_id_2(&q->sss, q->N_id_2);
  if (q->sss_generated) {
    const bool keep        = q->pss.chest_on_filter;
    q->pss.chest_on_filter = false;
    srsran_pss_filter(&q->pss, input, q->sss_recv);
    q->pss.chest_on_filter = keep;
    glue::Dot jobs[2] = {{q->sss_signal[0], q->sss_recv, (int)q->fft_size, glue::CONJ},
                         {q->sss_signal[1], q->sss_recv, (int)q->fft_size, glue::CONJ}};
    cf_t      r[2];
    if (glue::dots(jobs, 2, r)) {
      return false;
    }
    const float res0 = std::abs(r[0]), res1 = std::abs(r[1]);
    float       ratio;
    if (res0 > res1) {
      *sf_idx = 0;
      ratio   = res0 / res1;
    } else {
      *sf_idx = 5;
      ratio   = res1 / res0;
    }
    *N_id_1 = q->N_id_1;
    *corr   = ratio;
    return ratio > 1.2;
  }
  switch (q->sss_alg) {
    case SSS_DIFF:
      srsran_sss_m0m1_diff(&q->sss, input, &q->m0, &q->m0_value, &q->m1, &q->m1_value);
      break;
    case SSS_PARTIAL_3:
      srsran_sss_m0m1_partial(&q->sss, input, 3, NULL, &q->m0, &q->m0_value, &q->m1, &q->m1_value);
      break;
    case SSS_FULL:
      srsran_sss_m0m1_partial(&q->sss, input, 1, NULL, &q->m0, &q->m0_value, &q->m1, &q->m1_value);
      break;
  }
  *corr   = q->m0_value + q->m1_value;
  *sf_idx = srsran_sss_subframe(q->m0, q->m1);
  int ret = srsran_sss_N_id_1(&q->sss, q->m0, q->m1, *corr);
  if (ret >= 0) {
    *N_id_1 = (uint32_t)ret;
    return true;
  }
  return false;
}

// stage 1 (sync.c:592-622,663-675): PSS trials at -1 / 0 / +1 sub-carrier, keep the strongest
int stage_integer_cfo(srsran_sync_t* q, const cf_t** sig, uint32_t find_offset, int* peak_pos)
{
  srsran_pss_t* objs[3] = {&q->pss_i[0], &q->pss, &q->pss_i[1]};
  float         best    = -99;
  int           best_i  = 0;
  for (int t = 0; t < 3; t++) {
    float pv;
    srsran_pss_set_N_id_2(objs[t], q->N_id_2);
    int p = srsran_pss_find_pss(objs[t], &(*sig)[find_offset], &pv);
    if (p < 0) {
      return -1;
    }
    if (pv > best) {
      best          = pv;
      *peak_pos     = p;
      q->peak_value = pv;
      best_i        = t - 1;
    }
  }
  q->cfo_i_value = best_i;
  if (best_i != 0) {
    if (glue::prod(*sig, q->cfo_i_corr[best_i < 0 ? 0 : 1], q->temp, (int)q->frame_size, false)) {
      return -1;
    }
    *sig = q->temp;
  }
  return 0;
}

// stage 2 (sync.c:576-590,680-697): fractional CFO from the cyclic prefix, averaged, removed from the frame
int stage_cp_cfo(srsran_sync_t* q, const cf_t** sig)
{
  const uint32_t off = srsran_cp_synch(&q->cp_synch, *sig, q->max_offset, q->cfo_cp_nsymbols, (uint32_t)cp_samples(q->fft_size, 144));
  const cf_t     c   = srsran_cp_synch_corr_output(&q->cp_synch, off);
  const float    cfo = -std::arg(c) / ((float)M_PI * 2.0f);
  if (!q->cfo_cp_is_set) {
    q->cfo_cp_mean   = cfo;
    q->cfo_cp_is_set = true;
  } else {
    q->cfo_cp_mean = ema(cfo, q->cfo_cp_mean, q->cfo_ema_alpha);
  }
  srsran_cfo_correct(&q->cfo_corr_frame, *sig, q->temp, -q->cfo_cp_mean / q->fft_size);
  *sig = q->temp;
  return 0;
}

// stage 4a (sync.c:725-747): CFO from the two halves of the received PSS
void stage_pss_cfo(srsran_sync_t* q, const cf_t* sig, uint32_t find_offset, int peak_pos)
{
  const cf_t* pss_ptr = &sig[find_offset + peak_pos - q->fft_size];
  if (q->pss_filtering_enabled) {
    srsran_pss_filter(&q->pss, pss_ptr, q->pss_filt);
    pss_ptr = q->pss_filt;
  }
  q->cfo_pss = srsran_pss_cfo_compute(&q->pss, pss_ptr);
  if (!q->cfo_pss_is_set) {
    q->cfo_pss_mean   = q->cfo_pss;
    q->cfo_pss_is_set = true;
  } else if (15000 * fabsf(q->cfo_pss) < kMaxCfoPssOffsetHz) {
    q->cfo_pss_mean = ema(q->cfo_pss, q->cfo_pss_mean, q->cfo_ema_alpha);
  }
}

// stage 4b (sync.c:752-824): SSS for the FDD and/or TDD position
void stage_sss(srsran_sync_t* q, const cf_t* sig, uint32_t find_offset, int peak_pos)
{
  srsran_frame_type_t trials[2] = {SRSRAN_FDD, SRSRAN_TDD};
  uint32_t            ntrials   = 2;
  if (!q->detect_frame_type) {
    trials[0] = q->frame_type;
    ntrials   = 1;
  }
  float    corr[2] = {0, 0};
  uint32_t sf[2] = {0, 0}, nid1[2] = {0, 0};
  const int sym_sz = (int)q->fft_size + (q->cp == SRSRAN_CP_NORM ? cp_samples(q->fft_size, 144) : cp_samples(q->fft_size, 512));
  const int cp_sz  = sym_sz - (int)q->fft_size;
  q->sss_available = true;
  q->sss_detected  = false;
  for (uint32_t f = 0; f < ntrials; f++) {
    const int back    = trials[f] == SRSRAN_FDD ? 2 : 4; // SSS sits 1 (FDD) or 3 (TDD) symbols before the PSS
    const int sss_idx = (int)find_offset + peak_pos - back * sym_sz + cp_sz;
    if (sss_idx < 0) {
      q->sss_available = false;
      continue;
    }
    const cf_t* sss_ptr = &sig[sss_idx];
    if (q->cfo_pss_enable) {
      srsran_cfo_correct(&q->cfo_corr_symbol, sss_ptr, q->sss_filt, -q->cfo_pss_mean / q->fft_size);
      if (q->sss_channel_equalize && q->pss.chest_on_filter && q->pss_filtering_enabled) {
        cf_t* mid = &q->sss_filt[q->fft_size / 2 - SRSRAN_PSS_LEN / 2];
        glue::prod(mid, q->pss.tmp_ce, mid, SRSRAN_PSS_LEN, false);
      }
      sss_ptr = q->sss_filt;
    }
    q->sss_detected |= sss_of_symbol(q, sss_ptr, &sf[f], &nid1[f], &corr[f]);
  }
  if (q->detect_frame_type) {
    const int w   = corr[0] > corr[1] ? 0 : 1;
    q->frame_type = w == 0 ? SRSRAN_FDD : SRSRAN_TDD;
    q->sf_idx     = sf[w] + (w == 0 ? 0 : 1);
    q->N_id_1     = nid1[w];
    q->sss_corr   = corr[w];
  } else if (q->sss_detected) {
    q->sf_idx   = q->frame_type == SRSRAN_FDD ? sf[0] : sf[0] + 1;
    q->N_id_1   = nid1[0];
    q->sss_corr = corr[0];
  }
}

} // namespace

extern "C" srsran_sync_find_ret_t srsran_sync_find(srsran_sync_t* q, const cf_t* input, uint32_t find_offset, uint32_t* peak_position)
{
  if (!q) {
    return (srsran_sync_find_ret_t)SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (input == NULL || q->N_id_2 >= 3 || !fft_size_ok(q->fft_size)) {
    if (q->N_id_2 >= 3) {
      fprintf(stderr, "Must call srsran_sync_set_N_id_2() first!\n");
    }
    return SRSRAN_SYNC_ERROR;
  }
  q->sss_detected = false;
  if (peak_position) {
    *peak_position = 0;
  }
  const cf_t* sig      = input; // never modified: corrections go to q->temp
  int         peak_pos = 0;

  if (q->cfo_i_enable && stage_integer_cfo(q, &sig, find_offset, &peak_pos) < 0) {
    fprintf(stderr, "Error calling finding PSS sequence at : %d  \n", peak_pos);
    return SRSRAN_SYNC_ERROR;
  }
  if (q->cfo_cp_enable) {
    stage_cp_cfo(q, &sig);
  }
  if (!q->cfo_i_enable) { // with the integer stage the correlation has been done already
    srsran_pss_set_N_id_2(&q->pss, q->N_id_2);
    peak_pos = srsran_pss_find_pss(&q->pss, &sig[find_offset], q->threshold > 0 ? &q->peak_value : NULL);
    if (peak_pos < 0) {
      fprintf(stderr, "Error calling finding PSS sequence at : %d  \n", peak_pos);
      return SRSRAN_SYNC_ERROR;
    }
  }
  if (peak_position) {
    *peak_position = (uint32_t)peak_pos;
  }
  if (q->decimate && peak_pos < 0) {
    peak_pos = 0;
  }
  if (!(q->peak_value >= q->threshold || q->threshold == 0)) {
    return SRSRAN_SYNC_NOFOUND;
  }
  if (q->cfo_pss_enable && peak_pos >= (int)q->fft_size) {
    stage_pss_cfo(q, sig, find_offset, peak_pos);
  }
  // room for the SSS symbol and the CP measurement before the peak? (sync.c:750)
  if (peak_pos + find_offset < 2 * (q->fft_size + (uint32_t)cp_samples(q->fft_size, 512))) {
    return SRSRAN_SYNC_FOUND_NOSPACE;
  }
  if (q->sss_en) {
    stage_sss(q, sig, find_offset, peak_pos);
  }
  if (q->detect_cp) {
    srsran_sync_set_cp(q, srsran_sync_detect_cp(q, sig, peak_pos + find_offset));
  }
  return SRSRAN_SYNC_FOUND;
}
