// sync_host.cpp -- host side of the PSS / SSS search: replica + table generation, engines, handle ABI,
// batched cell search.
//
// Mirrors (interface + behaviour) lib/src/phy/sync/pss.c, sss.c, find_sss.c, gen_sss.c and the default
// path of srsran_sync_find (sync.c:629-843) of the reference.
#include "hip_common.h"
#include "srsran_amd/phy_sync_abi.h"
#include "sync_device.h"
#include "sync_glue.h"

#include <cmath>
#include <complex>
#include <vector>

using namespace phyhip;
typedef std::complex<double> cd;

static int cp_len_of(uint32_t symbol_sz, int c)
{
  return (int)ceilf((((float)(c) * (symbol_sz)) / 2048.0f)); // SRSRAN_CP_LEN, phy_common.h:125
}

// ------------------------------------------------------------------------------------------------ sequences

extern "C" int srsran_pss_generate(cf_t* signal, uint32_t N_id_2)
{
  // pss.c:341-368: Zadoff-Chu roots 25/29/34, length 63 with the DC element punctured, float arithmetic
  const float root_value[] = {25.0, 29.0, 34.0};
  if (N_id_2 > 2) {
    fprintf(stderr, "Invalid N_id_2 %d\n", N_id_2);
    return -1;
  }
  const int sign = -1;
  for (int i = 0; i < SRSRAN_PSS_LEN / 2; i++) {
    float arg = (float)sign * M_PI * root_value[N_id_2] * ((float)i * ((float)i + 1.0)) / 63.0;
    signal[i] = cf_t(cosf(arg), sinf(arg));
  }
  for (int i = SRSRAN_PSS_LEN / 2; i < SRSRAN_PSS_LEN; i++) {
    float arg = (float)sign * M_PI * root_value[N_id_2] * (((float)i + 2.0) * ((float)i + 1.0)) / 63.0;
    signal[i] = cf_t(cosf(arg), sinf(arg));
  }
  return 0;
}

extern "C" void srsran_pss_put_slot(cf_t* pss_signal, cf_t* slot, uint32_t nof_prb, srsran_cp_t cp)
{
  const int nsym = cp == SRSRAN_CP_NORM ? 7 : 6;
  int       k    = (nsym - 1) * nof_prb * 12 + nof_prb * 12 / 2 - 31;
  memset((void*)&slot[k - 5], 0, 5 * sizeof(cf_t));
  memcpy((void*)&slot[k], pss_signal, SRSRAN_PSS_LEN * sizeof(cf_t));
  memset((void*)&slot[k + SRSRAN_PSS_LEN], 0, 5 * sizeof(cf_t));
}

extern "C" void srsran_pss_get_slot(cf_t* slot, cf_t* pss_signal, uint32_t nof_prb, srsran_cp_t cp)
{
  const int nsym = cp == SRSRAN_CP_NORM ? 7 : 6;
  int       k    = (nsym - 1) * nof_prb * 12 + nof_prb * 12 / 2 - 31;
  memcpy((void*)pss_signal, &slot[k], SRSRAN_PSS_LEN * sizeof(cf_t));
}

namespace {

// gen_sss.c:31-53: the three m-sequences
void zsc_tilde(int* z_tilde, int* s_tilde, int* c_tilde)
{
  int x[SRSRAN_SSS_N];
  memset(x, 0, sizeof(x));
  x[4] = 1;
  for (int i = 0; i < 26; i++) {
    x[i + 5] = (x[i + 2] + x[i]) % 2;
  }
  for (int i = 0; i < SRSRAN_SSS_N; i++) {
    s_tilde[i] = 1 - 2 * x[i];
  }
  for (int i = 0; i < 26; i++) {
    x[i + 5] = (x[i + 3] + x[i]) % 2;
  }
  for (int i = 0; i < SRSRAN_SSS_N; i++) {
    c_tilde[i] = 1 - 2 * x[i];
  }
  for (int i = 0; i < 26; i++) {
    x[i + 5] = (x[i + 4] + x[i + 2] + x[i + 1] + x[i]) % 2;
  }
  for (int i = 0; i < SRSRAN_SSS_N; i++) {
    z_tilde[i] = 1 - 2 * x[i];
  }
}

void m0m1_of(uint32_t N_id_1, uint32_t* m0, uint32_t* m1)
{
  uint32_t q_prime = N_id_1 / (SRSRAN_SSS_N - 1);
  uint32_t q       = (N_id_1 + (q_prime * (q_prime + 1) / 2)) / (SRSRAN_SSS_N - 1);
  uint32_t m_prime = N_id_1 + (q * (q + 1) / 2);
  *m0              = m_prime % SRSRAN_SSS_N;
  *m1              = (*m0 + m_prime / SRSRAN_SSS_N + 1) % SRSRAN_SSS_N;
}

// time-domain replica of pss.c:31-62 evaluated in double: conj(IDFT_{mirror,dc,norm}(padded ZC)) / 62
void pss_time_replica(uint32_t N_id_2, uint32_t N, int cfo_i, cf_t* freq62, std::vector<cf_t>& time)
{
  srsran_pss_generate(freq62, N_id_2);
  std::vector<cd> pad(N, cd(0, 0)), dst(N, cd(0, 0));
  for (int j = 0; j < SRSRAN_PSS_LEN; j++) {
    pad[(N - SRSRAN_PSS_LEN) / 2 + cfo_i + j] = cd(freq62[j].real(), freq62[j].imag());
  }
  // copy_pre for a backward transform with mirror + dc (dft_fftw.c:297-308)
  const uint32_t hlen = N / 2;
  for (uint32_t i = 0; i + hlen + 1 < N + 0 && i < N - hlen - 1; i++) {
    dst[1 + i] = pad[hlen + i];
  }
  for (uint32_t i = 0; i < hlen; i++) {
    dst[N - hlen + i] = pad[i];
  }
  time.assign(N, cf_t(0, 0));
  const double norm = 1.0 / sqrt((double)N);
  // only <= 63 bins are non-zero: direct sum
  std::vector<uint32_t> nz;
  for (uint32_t k = 0; k < N; k++) {
    if (dst[k] != cd(0, 0)) {
      nz.push_back(k);
    }
  }
  for (uint32_t n = 0; n < N; n++) {
    cd acc(0, 0);
    for (uint32_t k : nz) {
      double a = 2.0 * M_PI * (double)((uint64_t)k * n % N) / (double)N;
      acc += dst[k] * cd(cos(a), sin(a));
    }
    acc *= norm;
    // conj, then * (float)(1/62) as srsran_vec_sc_prod_cfc
    float re = (float)acc.real(), im = -(float)acc.imag();
    float sc = (float)(1.0 / SRSRAN_PSS_LEN);
    time[n]  = cf_t(re * sc, im * sc);
  }
}

struct PssEngine {
  uint32_t frame_size = 0, fft_size = 0, max_caps = 0;
  int      n_blocks = 0, hop = 0, n_out = 0, block_n = 4096;
  size_t   corr_stride = 0;
  float2*  d_tw = nullptr;
  float2*  d_filt = nullptr;
  float2*  d_rep  = nullptr; // direct mode (frame_size < fft_size): the three time replicas
  bool     direct = false;
  float*   d_corr = nullptr;
  float*   d_part_val = nullptr;
  int*     d_part_idx = nullptr;
  float2*  d_spec = nullptr; // block spectra between the hypotheses of a launch (32 KB per block and capture)
  sync::PssResult* d_res = nullptr;
  cf_t     freq[3][SRSRAN_PSS_LEN];
  std::vector<cf_t> time[3];
};

void pss_engine_free(PssEngine* e)
{
  if (!e) {
    return;
  }
  (void)hipFree(e->d_tw);
  (void)hipFree(e->d_filt);
  (void)hipFree(e->d_rep);
  (void)hipFree(e->d_corr);
  (void)hipFree(e->d_part_val);
  (void)hipFree(e->d_part_idx);
  (void)hipFree(e->d_spec);
  (void)hipFree(e->d_res);
  delete e;
}

PssEngine* pss_engine_new(uint32_t frame_size, uint32_t fft_size, int cfo_i, uint32_t max_caps)
{
  if (!device_available()) {
    return nullptr;
  }
  if (fft_size > 2048 || fft_size < 64 || frame_size < 2) {
    set_error("PSS: unsupported sizes frame=%u fft=%u (64 <= fft <= 2048)", frame_size, fft_size);
    return nullptr;
  }
  auto* e        = new PssEngine;
  e->frame_size  = frame_size;
  e->fft_size    = fft_size;
  e->max_caps    = max_caps;
  e->direct      = frame_size < fft_size; // pss.c:476-481: sliding dot product, conv_output_len = frame_size
  e->n_out       = e->direct ? (int)frame_size - 1 : (int)(frame_size + fft_size) - 2; // pss.c:493: conv_output_len - 1 entries
  // overlap-save block of 4096 points.  Measured dead ends for long captures (a 2048-tap replica keeps only 2049 of 4096 outputs):
  // 8192-point blocks (6145 kept; spectrum + exchange image = 135 KB of LDS, one 512-lane workgroup per CU) run the 256-capture
  // bench in 1.92 ms against 1.97 ms -- the kernel is bound by barrier / LDS latency at the 2 waves per SIMD the LDS footprint
  // allows, not by FLOPs; 16384-point blocks need the spectrum in registers and spill (109 VGPRs at 1024 lanes, 232 at 512).
  e->block_n     = 4096;
  const int BN   = e->block_n;
  e->hop         = BN - (int)fft_size;
  e->n_blocks    = e->direct ? (e->n_out + 255) / 256 : (e->n_out + e->hop - 1) / e->hop;
  e->corr_stride = ((size_t)frame_size + fft_size + 2 + 3) & ~(size_t)3;
  // twiddles: W^i (i < BN), then the per-lane tables of pss_wave_kernel in [entry][lane] order (W^(lane j), W^(8 lane j), j < 8: coalesced loads)
  std::vector<std::complex<float>> tw(BN + 16 * 64), filt(3 * (size_t)BN);
  for (int i = 0; i < BN; i++) {
    double a = -2.0 * M_PI * (double)i / (double)BN;
    tw[i]    = std::complex<float>((float)cos(a), (float)sin(a));
  }
  for (int j = 0; j < 8; j++) {
    for (int l = 0; l < 64; l++) {
      tw[BN + j * 64 + l]       = tw[(l * j) % BN];
      tw[BN + (8 + j) * 64 + l] = tw[(l * 8 * j) % BN];
    }
  }
  for (uint32_t h = 0; h < 3; h++) {
    pss_time_replica(h, fft_size, cfo_i, e->freq[h], e->time[h]);
    if (e->direct) {
      continue;
    }
    // DFT_BN of the zero-padded replica, scaled by 1 / BN (double-precision radix-2 transform on the host)
    std::vector<cd> a(BN, cd(0, 0));
    for (uint32_t n = 0; n < fft_size; n++) {
      a[n] = cd(e->time[h][n].real(), e->time[h][n].imag());
    }
    for (int i = 1, j = 0; i < BN; i++) { // bit reversal
      int bit = BN >> 1;
      for (; j & bit; bit >>= 1) {
        j ^= bit;
      }
      j ^= bit;
      if (i < j) {
        std::swap(a[i], a[j]);
      }
    }
    for (int len = 2; len <= BN; len <<= 1) {
      for (int i = 0; i < BN; i += len) {
        for (int k = 0; k < len / 2; k++) {
          const double ang = -2.0 * M_PI * (double)k / (double)len;
          const cd     w(cos(ang), sin(ang));
          const cd     u = a[i + k], v = a[i + k + len / 2] * w;
          a[i + k]           = u + v;
          a[i + k + len / 2] = u - v;
        }
      }
    }
    for (int k = 0; k < BN; k++) {
      const cd v = a[k] / (double)BN;
      filt[(size_t)h * BN + k] = std::complex<float>((float)v.real(), (float)v.imag());
    }
  }
  bool ok = hipMalloc(&e->d_tw, tw.size() * sizeof(float2)) == hipSuccess &&
            hipMalloc(&e->d_filt, 3 * (size_t)BN * sizeof(float2)) == hipSuccess &&
            hipMalloc(&e->d_corr, (size_t)max_caps * 3 * e->corr_stride * sizeof(float)) == hipSuccess &&
            hipMalloc(&e->d_part_val, (size_t)max_caps * 3 * e->n_blocks * sizeof(float)) == hipSuccess &&
            hipMalloc(&e->d_part_idx, (size_t)max_caps * 3 * e->n_blocks * sizeof(int)) == hipSuccess &&
            hipMalloc(&e->d_spec, (size_t)max_caps * e->n_blocks * e->block_n * sizeof(float2)) == hipSuccess &&
            hipMalloc(&e->d_res, (size_t)max_caps * 3 * sizeof(sync::PssResult)) == hipSuccess &&
            upload(e->d_tw, tw.data(), tw.size() * sizeof(float2)) == hipSuccess &&
            upload(e->d_filt, filt.data(), 3 * (size_t)BN * sizeof(float2)) == hipSuccess &&
            hipMemset(e->d_corr, 0, (size_t)max_caps * 3 * e->corr_stride * sizeof(float)) == hipSuccess;
  if (ok && e->direct) {
    ok = hipMalloc(&e->d_rep, 3 * (size_t)fft_size * sizeof(float2)) == hipSuccess;
    for (uint32_t h = 0; h < 3 && ok; h++) {
      ok = upload(e->d_rep + (size_t)h * fft_size, e->time[h].data(), fft_size * sizeof(float2)) == hipSuccess;
    }
  }
  if (!ok) {
    set_error("PSS: device allocation failed");
    pss_engine_free(e);
    return nullptr;
  }
  return e;
}

// hypothesis slot 0 of the launch is bound to filter `first_filter` (the handle API searches ONE N_id_2 and
// shares one averaging buffer between them, like the reference)
int pss_engine_run(PssEngine* e, const void* d_in, uint32_t n_cap, int mask, int first_filter, float ema_alpha, hipStream_t st)
{
  sync::PssParams p;
  p.in          = d_in;
  p.twiddle     = e->d_tw;
  p.filt        = e->d_filt + (size_t)first_filter * e->block_n;
  p.corr        = e->d_corr;
  p.part_val    = e->d_part_val;
  p.part_idx    = e->d_part_idx;
  p.spec        = e->d_spec;
  p.in_stride   = e->frame_size;
  p.corr_stride = e->corr_stride;
  p.n_cap       = (int)n_cap;
  p.n_blocks    = e->n_blocks;
  p.hop         = e->hop;
  p.part_span   = e->direct ? 256 : e->hop;
  p.fft_size    = (int)e->fft_size;
  p.frame_size  = (int)e->frame_size;
  p.n_out       = e->n_out;
  p.n_id_2_mask = mask;
  p.ema_alpha   = ema_alpha;
  if (e->direct) {
    if (n_cap != 1 || mask != 1) {
      set_error("PSS: the sliding dot-product mode (frame < fft) serves one capture and one N_id_2");
      return SRSRAN_ERROR;
    }
    PHY_HIP_CHECK(sync::launch_pss_direct(p, e->d_rep + (size_t)first_filter * e->fft_size, e->d_res, st), SRSRAN_ERROR);
    return SRSRAN_SUCCESS;
  }
  PHY_HIP_CHECK(sync::launch_pss(p, e->d_res, st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

struct SssEngine {
  uint32_t fft_size = 0;
  float2*  d_tw = nullptr;
  float*   d_seq = nullptr; // s_tilde, c_tilde, z_tilde (3 x 31)
};

void sss_engine_free(SssEngine* e)
{
  if (!e) {
    return;
  }
  (void)hipFree(e->d_tw);
  (void)hipFree(e->d_seq);
  delete e;
}

SssEngine* sss_engine_new(uint32_t fft_size)
{
  if (!device_available()) {
    return nullptr;
  }
  if (fft_size < 64 || fft_size > 2048 || fft_size % 64) { // the symbol transform is N / 64 interleaved 64-point transforms (sss_kernel)
    set_error("SSS: unsupported symbol size %u (a multiple of 64 up to 2048)", fft_size);
    return nullptr;
  }
  auto* e     = new SssEngine;
  e->fft_size = fft_size;
  std::vector<std::complex<float>> tw(fft_size);
  for (uint32_t i = 0; i < fft_size; i++) {
    double a = -2.0 * M_PI * (double)i / (double)fft_size;
    tw[i]    = std::complex<float>((float)cos(a), (float)sin(a));
  }
  int   zt[31], st[31], ct[31];
  float seq[93];
  zsc_tilde(zt, st, ct);
  for (int i = 0; i < 31; i++) {
    seq[i]      = (float)st[i];
    seq[31 + i] = (float)ct[i];
    seq[62 + i] = (float)zt[i];
  }
  bool ok = hipMalloc(&e->d_tw, fft_size * sizeof(float2)) == hipSuccess && hipMalloc(&e->d_seq, sizeof(seq)) == hipSuccess &&
            upload(e->d_tw, tw.data(), fft_size * sizeof(float2)) == hipSuccess &&
            upload(e->d_seq, seq, sizeof(seq)) == hipSuccess;
  if (!ok) {
    set_error("SSS: device allocation failed");
    sss_engine_free(e);
    return nullptr;
  }
  return e;
}

void sss_params(sync::SssParams* p, const SssEngine* e, const void* d_in, size_t in_stride, uint32_t n_cap, uint32_t frame_size,
                srsran_cp_t cp, int M, int mask, float threshold, const int* d_pos, const void* d_ce)
{
  p->in          = d_in;
  p->twiddle     = e->d_tw;
  p->s_tilde     = e->d_seq;
  p->c_tilde     = e->d_seq + 31;
  p->z_tilde     = e->d_seq + 62;
  p->sss_pos     = d_pos;
  p->ce          = d_ce;
  p->in_stride   = in_stride;
  p->n_cap       = (int)n_cap;
  p->fft_size    = (int)e->fft_size;
  p->frame_size  = (int)frame_size;
  p->cp_len      = cp == SRSRAN_CP_NORM ? cp_len_of(e->fft_size, 144) : cp_len_of(e->fft_size, 512);
  p->cp_ext_len  = cp_len_of(e->fft_size, 512);
  p->M           = M;
  p->n_id_2_mask = mask;
  p->threshold   = threshold;
}

} // namespace

// ------------------------------------------------------------------------------------------------ PSS handle ABI

namespace {
struct PssCtx {
  DeviceTag tag;
  PssEngine*  e = nullptr;
  hipStream_t stream = nullptr;
  float2*     d_in = nullptr;
  float2*     d_dec = nullptr; // decimated + filtered input (decimate > 1)
  cf_t*       h_in = nullptr;
  float*      h_avg = nullptr;
  sync::PssResult* h_res = nullptr;
  int         cfo_i = 0;
  size_t      in_cap = 0;
  bool        plans = false;   // dftp_input / idftp_input created (srsran_pss_filter and friends)
};
PssCtx* pctx(srsran_pss_t* q)
{
  return reinterpret_cast<PssCtx*>(q->conv_fft.input_fft);
}
void pctx_free(PssCtx* c)
{
  if (!c) {
    return;
  }
  pss_engine_free(c->e);
  (void)hipFree(c->d_in);
  (void)hipFree(c->d_dec);
  (void)hipHostFree(c->h_in);
  (void)hipHostFree(c->h_avg);
  (void)hipHostFree(c->h_res);
  if (c->stream) {
    (void)hipStreamDestroy(c->stream);
  }
  delete c;
}
int pss_setup(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size, int offset)
{
  PssCtx* c = pctx(q);
  if (c->e) {
    pss_engine_free(c->e);
    c->e = nullptr;
  }
  c->e = pss_engine_new(frame_size, fft_size, offset, 1);
  if (!c->e) {
    fprintf(stderr, "[srsran_phy_hip] srsran_pss: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  c->cfo_i = offset;
  for (int h = 0; h < 3; h++) {
    memcpy((void*)q->pss_signal_freq[h], c->e->freq[h], sizeof(c->e->freq[h]));
    memset((void*)q->pss_signal_time[h], 0, sizeof(cf_t) * (q->max_fft_size + q->max_frame_size + 1));
    memcpy((void*)q->pss_signal_time[h], c->e->time[h].data(), sizeof(cf_t) * fft_size);
  }
  memset(q->conv_output_avg, 0, sizeof(float) * (q->max_fft_size + q->max_frame_size + 1));
  return SRSRAN_SUCCESS;
}
} // namespace

// 4-tap decimation low-pass of the reference (filter.c:27-62, the order-3 rows of its tables), per factor 2 / 3 / 4
static const float kDecimTaps[3][4] = {{0.0167364016736f, 0.48326359832636f, 0.48326359832636f, 0.01673640167364f},
                                       {0.032388663967611f, 0.467611336032389f, 0.467611336032389f, 0.032388663967611f},
                                       {0.038579006748772f, 0.461420993251228f, 0.461420993251228f, 0.038579006748772f}};

extern "C" int srsran_pss_init_fft_offset_decim(srsran_pss_t* q, uint32_t max_frame_size, uint32_t max_fft_size, int offset, int decimate)
{
  if (q == NULL) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  memset(q, 0, sizeof(srsran_pss_t));
  if (decimate < 1 || decimate > 4) {
    fprintf(stderr, "[srsran_phy_hip] srsran_pss: decimation factor %d not supported (1..4)\n", decimate);
    return SRSRAN_ERROR;
  }
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  q->N_id_2         = 10;
  q->ema_alpha      = 0.2;
  q->max_fft_size   = max_fft_size;
  q->max_frame_size = max_frame_size;
  q->decimate       = decimate;
  q->fft_size       = max_fft_size / decimate; // pss.c:107-111
  q->frame_size     = max_frame_size / decimate;
  const size_t buffer_size = (size_t)max_fft_size + max_frame_size + 1;
  auto* c = new PssCtx;
  q->conv_fft.input_fft = reinterpret_cast<cf_t*>(c);
  q->conv_output_avg    = (float*)calloc(buffer_size, sizeof(float));
  q->conv_output_abs    = (float*)calloc(buffer_size, sizeof(float));
  for (int h = 0; h < 3; h++) {
    q->pss_signal_time[h] = (cf_t*)calloc(buffer_size, sizeof(cf_t));
  }
  if (decimate > 1) {
    q->filter.factor       = decimate;
    q->filter.num_taps     = 4; // filter_order 3, pss.c:121
    q->filter.is_decimator = true;
    q->filter.taps         = (float*)malloc(4 * sizeof(float));
    if (q->filter.taps) {
      memcpy(q->filter.taps, kDecimTaps[decimate - 2], 4 * sizeof(float));
    }
  }
  // full-rate input: frame (+ fft in the sliding dot-product mode)
  c->in_cap = (size_t)max_frame_size + max_fft_size;
  bool ok = q->conv_output_avg && q->conv_output_abs && q->pss_signal_time[0] && q->pss_signal_time[1] && q->pss_signal_time[2] &&
            (decimate == 1 || q->filter.taps) &&
            hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc(&c->d_in, c->in_cap * sizeof(float2)) == hipSuccess &&
            hipMalloc(&c->d_dec, c->in_cap * sizeof(float2)) == hipSuccess &&
            hipHostMalloc(&c->h_in, c->in_cap * sizeof(cf_t)) == hipSuccess &&
            hipHostMalloc(&c->h_avg, buffer_size * sizeof(float)) == hipSuccess &&
            hipHostMalloc(&c->h_res, 3 * sizeof(sync::PssResult)) == hipSuccess;
  if (!ok || pss_setup(q, q->frame_size, q->fft_size, offset)) {
    srsran_pss_free(q);
    return SRSRAN_ERROR;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_pss_init_fft_offset(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size, int offset)
{
  return srsran_pss_init_fft_offset_decim(q, frame_size, fft_size, offset, 1);
}

extern "C" int srsran_pss_init_fft(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size)
{
  return srsran_pss_init_fft_offset(q, frame_size, fft_size, 0);
}

extern "C" int srsran_pss_init(srsran_pss_t* q, uint32_t frame_size)
{
  return srsran_pss_init_fft(q, frame_size, 128);
}

extern "C" int srsran_pss_resize(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size, int offset)
{
  if (q == NULL) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (fft_size > q->max_fft_size || frame_size > q->max_frame_size) {
    fprintf(stderr, "Error in pss_config(): fft_size and frame_size must be lower than initialized\n");
    return SRSRAN_ERROR;
  }
  q->N_id_2     = 10;
  q->ema_alpha  = 0.2;
  q->fft_size   = fft_size / q->decimate; // pss.c:237-241
  q->frame_size = frame_size / q->decimate;
  PssCtx* c = pctx(q);
  if (c && c->plans) {
    if (srsran_dft_replan(&q->dftp_input, (int)q->fft_size) || srsran_dft_replan(&q->idftp_input, (int)q->fft_size)) {
      return SRSRAN_ERROR;
    }
  }
  memset((void*)q->tmp_fft2, 0, sizeof(q->tmp_fft2));
  return pss_setup(q, q->frame_size, q->fft_size, offset);
}

extern "C" void srsran_pss_free(srsran_pss_t* q)
{
  if (!q) {
    return;
  }
  PssCtx* c = pctx(q);
  if (c && c->plans) {
    srsran_dft_plan_free(&q->dftp_input);
    srsran_dft_plan_free(&q->idftp_input);
  }
  pctx_free(c);
  for (int h = 0; h < 3; h++) {
    free(q->pss_signal_time[h]);
  }
  free(q->conv_output_avg);
  free(q->conv_output_abs);
  free(q->filter.taps);
  memset(q, 0, sizeof(srsran_pss_t));
}

extern "C" void srsran_pss_reset(srsran_pss_t* q)
{
  PssCtx* c = pctx(q);
  memset(q->conv_output_avg, 0, sizeof(float) * (q->fft_size + q->frame_size + 1));
  if (c && c->e) {
    (void)hipMemsetAsync(c->e->d_corr, 0, c->e->corr_stride * sizeof(float), c->stream);
    (void)hipStreamSynchronize(c->stream);
  }
}

extern "C" int srsran_pss_set_N_id_2(srsran_pss_t* q, uint32_t N_id_2)
{
  if (N_id_2 > 2) {
    fprintf(stderr, "Invalid N_id_2 %d\n", N_id_2);
    return -1;
  }
  q->N_id_2 = N_id_2;
  return 0;
}

extern "C" void srsran_pss_set_ema_alpha(srsran_pss_t* q, float alpha)
{
  q->ema_alpha = alpha;
}

extern "C" int srsran_pss_find_pss(srsran_pss_t* q, const cf_t* input, float* corr_peak_value)
{
  if (q == NULL || input == NULL) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (q->N_id_2 > 2) {
    fprintf(stderr, "Error finding PSS peak, Must set N_id_2 first\n");
    return SRSRAN_ERROR;
  }
  PssCtx* c = pctx(q);
  if (!c || !c->e) {
    return SRSRAN_ERROR;
  }
  PHY_DEV_GUARD(c->tag, "srsran_pss_find_pss", SRSRAN_ERROR);
  const bool direct = q->frame_size < q->fft_size;
  // samples the reference reads: frame_size * decimate (pss.c:462), or frame_size + fft_size - 1 in the
  // sliding dot-product mode (pss.c:477-479)
  const size_t n_in = direct ? (size_t)q->frame_size + q->fft_size - 1 : (size_t)q->frame_size * q->decimate;
  if (n_in > c->in_cap) {
    return SRSRAN_ERROR;
  }
  memcpy(c->h_in, input, n_in * sizeof(cf_t));
  PHY_HIP_CHECK(hipMemcpyAsync(c->d_in, c->h_in, n_in * sizeof(cf_t), hipMemcpyHostToDevice, c->stream), SRSRAN_ERROR);
  const float2* d_sig = c->d_in;
  if (q->decimate > 1 && !direct) {
    PHY_HIP_CHECK(sync::launch_decim(c->d_in, c->d_dec, (int)q->frame_size, q->decimate, q->filter.taps, c->stream), SRSRAN_ERROR);
    d_sig = c->d_dec;
  }
  if (pss_engine_run(c->e, d_sig, 1, 1, (int)q->N_id_2, q->ema_alpha, c->stream)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_pss_find_pss: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  const size_t n_avg = (size_t)c->e->n_out;
  PHY_HIP_CHECK(hipMemcpyAsync(c->h_res, c->e->d_res, sizeof(sync::PssResult), hipMemcpyDeviceToHost, c->stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(c->h_avg, c->e->d_corr, n_avg * sizeof(float), hipMemcpyDeviceToHost, c->stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(c->stream), SRSRAN_ERROR);
  memcpy(q->conv_output_avg, c->h_avg, n_avg * sizeof(float));
  q->peak_value = c->h_res->peak_value;
  if (corr_peak_value) {
    *corr_peak_value = c->h_res->psr; // SRSRAN_PSS_RETURN_PSR (pss.h:61)
  }
  uint32_t pos = (uint32_t)c->h_res->peak_pos;
  if (q->decimate > 1) {
    pos = (pos - (uint32_t)(q->filter.num_taps - 2)) * (uint32_t)q->decimate; // pss.c:521-525 (unsigned, as there)
  }
  return direct ? (int)pos + (int)q->fft_size : (int)pos;
}

// ---- PSS helpers of pss.c:536-640: channel estimate, central-band filter, half-symbol CFO, cancellation

static int pss_plans(srsran_pss_t* q)
{
  PssCtx* c = pctx(q);
  if (!c) {
    return SRSRAN_ERROR;
  }
  if (c->plans) {
    return SRSRAN_SUCCESS;
  }
  // pss.c:127-141: both plans mirror + dc, no normalisation
  if (srsran_dft_plan(&q->dftp_input, (int)q->fft_size, SRSRAN_DFT_FORWARD, SRSRAN_DFT_COMPLEX)) {
    return SRSRAN_ERROR;
  }
  if (srsran_dft_plan(&q->idftp_input, (int)q->fft_size, SRSRAN_DFT_BACKWARD, SRSRAN_DFT_COMPLEX)) {
    srsran_dft_plan_free(&q->dftp_input);
    return SRSRAN_ERROR;
  }
  for (srsran_dft_plan_t* pl : {&q->dftp_input, &q->idftp_input}) {
    srsran_dft_plan_set_mirror(pl, true);
    srsran_dft_plan_set_dc(pl, true);
    srsran_dft_plan_set_norm(pl, false);
  }
  c->plans = true;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_pss_filter_enable(srsran_pss_t* q, bool enable)
{
  q->filter_pss_enable = enable;
}

extern "C" int srsran_pss_chest(srsran_pss_t* q, const cf_t* input, cf_t ce[SRSRAN_PSS_LEN])
{
  if (q == NULL || input == NULL) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (q->N_id_2 > 2) {
    fprintf(stderr, "Error finding PSS peak, Must set N_id_2 first\n");
    return SRSRAN_ERROR;
  }
  if (pss_plans(q)) {
    return SRSRAN_ERROR;
  }
  srsran_dft_run_c(&q->dftp_input, input, q->tmp_fft);
  return glue::prod(&q->tmp_fft[(q->fft_size - SRSRAN_PSS_LEN) / 2], q->pss_signal_freq[q->N_id_2], ce, SRSRAN_PSS_LEN, true);
}

extern "C" void srsran_pss_filter(srsran_pss_t* q, const cf_t* input, cf_t* output)
{
  if (pss_plans(q)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_pss_filter: %s\n", get_error());
    return;
  }
  srsran_dft_run_c(&q->dftp_input, input, q->tmp_fft);
  // keep the 62 PSS sub-carriers; the rest of tmp_fft2 stays zero since init / resize (pss.c:143,262)
  memcpy(&q->tmp_fft2[q->fft_size / 2 - SRSRAN_PSS_LEN / 2], &q->tmp_fft[q->fft_size / 2 - SRSRAN_PSS_LEN / 2],
         sizeof(cf_t) * SRSRAN_PSS_LEN);
  if (q->chest_on_filter) {
    glue::prod(&q->tmp_fft[(q->fft_size - SRSRAN_PSS_LEN) / 2], q->pss_signal_freq[q->N_id_2], q->tmp_ce, SRSRAN_PSS_LEN, true);
  }
  srsran_dft_run_c(&q->idftp_input, q->tmp_fft2, output);
}

extern "C" float srsran_pss_cfo_compute(srsran_pss_t* q, const cf_t* pss_recv)
{
  const cf_t* ptr = pss_recv;
  if (q->filter_pss_enable) {
    srsran_pss_filter(q, pss_recv, q->tmp_fft);
    ptr = q->tmp_fft;
  }
  const int   half  = (int)q->fft_size / 2;
  const cf_t* rep   = q->pss_signal_time[q->N_id_2];
  glue::Dot   jobs[2] = {{rep, ptr, half, glue::PLAIN}, {&rep[half], &ptr[half], half, glue::PLAIN}};
  cf_t        y[2];
  if (glue::dots(jobs, 2, y)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_pss_cfo_compute: %s\n", get_error());
    return 0.f;
  }
  return std::arg(std::conj(y[0]) * y[1]) / (float)M_PI; // pss.c:639
}

extern "C" void srsran_pss_sic(srsran_pss_t* q, cf_t* input)
{
  if (!q->chest_on_filter) {
    fprintf(stderr, "Error calling srsran_pss_sic(): need to enable channel estimation on filtering\n");
    return;
  }
  if (pss_plans(q)) {
    return;
  }
  memset((void*)q->tmp_fft, 0, sizeof(cf_t) * q->fft_size);
  glue::prod(q->pss_signal_freq[q->N_id_2], q->tmp_ce, &q->tmp_fft[(q->fft_size - SRSRAN_PSS_LEN) / 2], SRSRAN_PSS_LEN, false);
  srsran_dft_run_c(&q->idftp_input, q->tmp_fft, q->tmp_fft2);
  // input -= received replica / fft_size; the scaled replica stays in tmp_fft2 (pss.c:552-553)
  glue::lincomb(q->tmp_fft2, 1.0f / (float)q->fft_size, q->tmp_fft2, 0.f, q->tmp_fft2, (int)q->fft_size);
  glue::lincomb(input, 1.0f, q->tmp_fft2, -1.0f, input, (int)q->fft_size);
}

// ------------------------------------------------------------------------------------------------ SSS handle ABI

namespace {
struct SssCtx {
  DeviceTag tag;
  SssEngine*  e = nullptr;
  hipStream_t stream = nullptr;
  float2*     d_in = nullptr;
  float2*     d_ce = nullptr;
  int*        d_pos = nullptr;
  sync::SssResult* d_res = nullptr;
  cf_t*       h_in = nullptr;
  sync::SssResult* h_res = nullptr;
};
SssCtx* sctx(srsran_sss_t* q)
{
  return reinterpret_cast<SssCtx*>(q->dftp_input.p);
}
} // namespace

extern "C" void srsran_sss_generate(float* signal0, float* signal5, uint32_t cell_id)
{
  // gen_sss.c:125-163
  uint32_t id1 = cell_id / 3, id2 = cell_id % 3, m0, m1;
  int      s_t[SRSRAN_SSS_N], c_t[SRSRAN_SSS_N], z_t[SRSRAN_SSS_N];
  m0m1_of(id1, &m0, &m1);
  zsc_tilde(z_t, s_t, c_t);
  for (int i = 0; i < SRSRAN_SSS_N; i++) {
    int s0 = s_t[(i + m0) % 31], s1 = s_t[(i + m1) % 31];
    int c0 = c_t[(i + id2) % 31], c1 = c_t[(i + id2 + 3) % 31];
    int z10 = z_t[(i + (m0 % 8)) % 31], z11 = z_t[(i + (m1 % 8)) % 31];
    signal0[2 * i]     = (float)(s0 * c0);
    signal0[2 * i + 1] = (float)(s1 * c1 * z10);
    signal5[2 * i]     = (float)(s1 * c0);
    signal5[2 * i + 1] = (float)(s0 * c1 * z11);
  }
}

extern "C" void srsran_sss_put_slot(float* sss, cf_t* slot, uint32_t nof_prb, srsran_cp_t cp)
{
  const uint32_t nsym = cp == SRSRAN_CP_NORM ? 7 : 6;
  uint32_t       k    = (nsym - 2) * nof_prb * 12 + nof_prb * 12 / 2 - 31;
  if (k > 5) {
    memset((void*)&slot[k - 5], 0, 5 * sizeof(cf_t));
    for (uint32_t i = 0; i < SRSRAN_SSS_LEN; i++) {
      slot[k + i] = cf_t(sss[i], 0);
    }
    memset((void*)&slot[k + SRSRAN_SSS_LEN], 0, 5 * sizeof(cf_t));
  }
}

extern "C" void srsran_sss_free(srsran_sss_t* q)
{
  SssCtx* c = sctx(q);
  if (c) {
    sss_engine_free(c->e);
    (void)hipFree(c->d_in);
    (void)hipFree(c->d_ce);
    (void)hipFree(c->d_pos);
    (void)hipFree(c->d_res);
    (void)hipHostFree(c->h_in);
    (void)hipHostFree(c->h_res);
    if (c->stream) {
      (void)hipStreamDestroy(c->stream);
    }
    delete c;
  }
  memset(q, 0, sizeof(srsran_sss_t));
}

extern "C" int srsran_sss_init(srsran_sss_t* q, uint32_t fft_size)
{
  if (q == NULL || fft_size > 2048) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  memset(q, 0, sizeof(srsran_sss_t));
  auto* c = new SssCtx;
  q->dftp_input.p = c;
  c->e            = sss_engine_new(fft_size);
  int zero        = 0;
  bool ok = c->e && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc(&c->d_in, (size_t)fft_size * sizeof(float2)) == hipSuccess &&
            hipMalloc(&c->d_ce, 3 * 62 * sizeof(float2)) == hipSuccess && hipMalloc(&c->d_pos, 3 * sizeof(int)) == hipSuccess &&
            hipMalloc(&c->d_res, 3 * sizeof(sync::SssResult)) == hipSuccess &&
            hipHostMalloc(&c->h_in, ((size_t)fft_size + 62) * sizeof(cf_t)) == hipSuccess &&
            hipHostMalloc(&c->h_res, 3 * sizeof(sync::SssResult)) == hipSuccess &&
            hipMemset(c->d_pos, 0, 3 * sizeof(int)) == hipSuccess;
  (void)zero;
  if (!ok) {
    fprintf(stderr, "[srsran_phy_hip] srsran_sss_init: %s\n", get_error());
    srsran_sss_free(q);
    return SRSRAN_ERROR;
  }
  q->dftp_input.size      = (int)fft_size;
  q->dftp_input.init_size = (int)fft_size;
  q->dftp_input.forward   = true;
  q->dftp_input.mirror    = true;
  q->dftp_input.dc        = true;
  q->fft_size             = fft_size;
  q->max_fft_size         = fft_size;
  // gen_sss.c:64-73 and find_sss.c:194-225
  for (uint32_t id = 0; id < 168; id++) {
    uint32_t m0, m1;
    m0m1_of(id, &m0, &m1);
    q->N_id_1_table[m0][m1 - 1] = id;
  }
  int zt[31], st[31], ct[31];
  zsc_tilde(zt, st, ct);
  for (uint32_t h = 0; h < 3; h++) {
    srsran_sss_fc_tables_t* t = &q->fc_tables[h];
    for (int i = 0; i < 31; i++) {
      for (int j = 0; j < 31; j++) {
        t->z1[i][j] = (float)zt[(j + (i % 8)) % 31];
        t->s[i][j]  = (float)st[(j + i) % 31];
      }
      for (int j = 0; j < 30; j++) {
        t->sd[i][j] = (float)(st[(j + 1 + i) % 31] * st[(j + i) % 31]);
      }
      t->c[0][i] = (float)ct[(i + h) % 31];
      t->c[1][i] = (float)ct[(i + h + 3) % 31];
    }
  }
  q->N_id_2 = 0;
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_sss_resize(srsran_sss_t* q, uint32_t fft_size)
{
  if (q == NULL || fft_size > 2048) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (fft_size > q->max_fft_size) {
    fprintf(stderr, "Error in sss_synch_resize(): fft_size must be lower than initialized\n");
    return SRSRAN_ERROR;
  }
  SssCtx* c = sctx(q);
  sss_engine_free(c->e);
  c->e = sss_engine_new(fft_size);
  if (!c->e) {
    return SRSRAN_ERROR;
  }
  q->fft_size        = fft_size;
  q->dftp_input.size = (int)fft_size;
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_sss_set_N_id_2(srsran_sss_t* q, uint32_t N_id_2)
{
  if (N_id_2 > 2) {
    fprintf(stderr, "Invalid N_id_2 %d\n", N_id_2);
    return SRSRAN_ERROR;
  }
  q->N_id_2 = N_id_2;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_sss_set_threshold(srsran_sss_t* q, float threshold)
{
  q->corr_peak_threshold = threshold;
}

extern "C" uint32_t srsran_sss_subframe(uint32_t m0, uint32_t m1)
{
  return m1 > m0 ? 0 : 5;
}

extern "C" int srsran_sss_N_id_1(srsran_sss_t* q, uint32_t m0, uint32_t m1, float corr)
{
  int N_id_1 = SRSRAN_ERROR;
  if (corr > q->corr_peak_threshold) { // sss.c:139-156
    if (m1 > m0) {
      if (m0 < 30 && m1 - 1 < 30) {
        N_id_1 = (int)q->N_id_1_table[m0][m1 - 1];
      }
    } else {
      if (m1 < 30 && m0 - 1 < 30) {
        N_id_1 = (int)q->N_id_1_table[m1][m0 - 1];
      }
    }
  }
  return N_id_1;
}

static int sss_run(srsran_sss_t* q, const cf_t* input, int M, cf_t* ce, uint32_t* m0, float* m0_value, uint32_t* m1, float* m1_value)
{
  if (q == NULL || input == NULL || m0 == NULL || m1 == NULL || M > 3) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  SssCtx* c = sctx(q);
  if (!c || !c->e) {
    return SRSRAN_ERROR;
  }
  PHY_DEV_GUARD(c->tag, "srsran_sss", SRSRAN_ERROR);
  const uint32_t N = q->fft_size;
  memcpy(c->h_in, input, N * sizeof(cf_t));
  PHY_HIP_CHECK(hipMemcpyAsync(c->d_in, c->h_in, N * sizeof(cf_t), hipMemcpyHostToDevice, c->stream), SRSRAN_ERROR);
  if (ce) {
    memcpy(c->h_in + N, ce, 62 * sizeof(cf_t));
    PHY_HIP_CHECK(hipMemcpyAsync(c->d_ce + (size_t)q->N_id_2 * 62, c->h_in + N, 62 * sizeof(cf_t), hipMemcpyHostToDevice, c->stream), SRSRAN_ERROR);
  }
  sync::SssParams p;
  sss_params(&p, c->e, c->d_in, N, 1, N, SRSRAN_CP_NORM, M, 1 << q->N_id_2, -1.0f, c->d_pos, ce ? c->d_ce : nullptr);
  PHY_HIP_CHECK(sync::launch_sss(p, nullptr, c->d_res, c->stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(c->h_res, c->d_res, 3 * sizeof(sync::SssResult), hipMemcpyDeviceToHost, c->stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(c->stream), SRSRAN_ERROR);
  const sync::SssResult& r = c->h_res[q->N_id_2];
  *m0 = r.m0;
  *m1 = r.m1;
  if (m0_value) {
    *m0_value = r.m0_value;
  }
  if (m1_value) {
    *m1_value = r.m1_value;
  }
  q->corr_output_m0[r.m0] = r.m0_value;
  q->corr_output_m1[r.m1] = r.m1_value;
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_sss_m0m1_partial(srsran_sss_t* q, const cf_t* input, uint32_t M, cf_t ce[2 * SRSRAN_SSS_N], uint32_t* m0,
                                       float* m0_value, uint32_t* m1, float* m1_value)
{
  if (M == 0) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  return sss_run(q, input, (int)M, ce, m0, m0_value, m1, m1_value);
}

extern "C" int srsran_sss_m0m1_diff_coh(srsran_sss_t* q, const cf_t* input, cf_t ce[2 * SRSRAN_SSS_N], uint32_t* m0, float* m0_value,
                                        uint32_t* m1, float* m1_value)
{
  return sss_run(q, input, 0, ce, m0, m0_value, m1, m1_value);
}

extern "C" int srsran_sss_m0m1_diff(srsran_sss_t* q, const cf_t* input, uint32_t* m0, float* m0_value, uint32_t* m1, float* m1_value)
{
  return srsran_sss_m0m1_diff_coh(q, input, NULL, m0, m0_value, m1, m1_value);
}

// ------------------------------------------------------------------------------------------------ batched cell search

struct srsran_hip_cellsearch {
  DeviceTag tag;
  PssEngine*  pss = nullptr;
  SssEngine*  sss = nullptr;
  sync::SssResult* d_sss = nullptr;
  srsran_cp_t cp = SRSRAN_CP_NORM;
  int         M = 1;
  uint32_t    max_caps = 0;
};

extern "C" int srsran_hip_cellsearch_create(srsran_hip_cellsearch_t** hh, uint32_t frame_size, uint32_t fft_size, srsran_cp_t cp,
                                            int sss_alg, uint32_t max_captures)
{
  if (!hh || max_captures == 0 || !(sss_alg == 0 || sss_alg == 1 || sss_alg == 3)) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh     = nullptr;
  auto* h = new srsran_hip_cellsearch;
  h->pss  = pss_engine_new(frame_size, fft_size, 0, max_captures);
  h->sss  = h->pss ? sss_engine_new(fft_size) : nullptr;
  if (!h->pss || !h->sss || hipMalloc(&h->d_sss, (size_t)max_captures * 3 * sizeof(sync::SssResult)) != hipSuccess) {
    srsran_hip_cellsearch_free(h);
    return SRSRAN_ERROR;
  }
  h->cp       = cp;
  h->M        = sss_alg;
  h->max_caps = max_captures;
  *hh         = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_cellsearch_free(srsran_hip_cellsearch_t* h)
{
  if (!h) {
    return;
  }
  pss_engine_free(h->pss);
  sss_engine_free(h->sss);
  (void)hipFree(h->d_sss);
  delete h;
}

extern "C" int srsran_hip_cellsearch_run(srsran_hip_cellsearch_t* h, const cf_t* d_captures, uint32_t n_captures, int n_id_2_mask,
                                         srsran_hip_cell_t* d_cells, void* stream)
{
  TraceRange trace_("srsran_hip_cellsearch_run");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_cellsearch_run", SRSRAN_ERROR);
  }
  if (!h || !d_captures || !d_cells || n_captures == 0 || n_captures > h->max_caps || !(n_id_2_mask & 7)) {
    set_error("cellsearch: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t st = (hipStream_t)stream;
  // ema_alpha = 1: stateless search (each capture on its own), as the batched use needs
  if (pss_engine_run(h->pss, d_captures, n_captures, n_id_2_mask & 7, 0, 1.0f, st)) {
    return SRSRAN_ERROR;
  }
  sync::SssParams p;
  sss_params(&p, h->sss, d_captures, h->pss->frame_size, n_captures, h->pss->frame_size, h->cp, h->M, n_id_2_mask & 7, 0.0f, nullptr,
             nullptr);
  PHY_HIP_CHECK(sync::launch_sss(p, h->pss->d_res, h->d_sss, st), SRSRAN_ERROR);
  static_assert(sizeof(sync::CellResult) == sizeof(srsran_hip_cell_t), "cell result layout");
  PHY_HIP_CHECK(sync::launch_pack(h->pss->d_res, h->d_sss, reinterpret_cast<sync::CellResult*>(d_cells), (int)n_captures * 3, st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" const float* srsran_hip_cellsearch_corr(srsran_hip_cellsearch_t* h, uint32_t capture, uint32_t N_id_2)
{
  if (!h || capture >= h->max_caps || N_id_2 > 2) {
    return nullptr;
  }
  return h->pss->d_corr + ((size_t)capture * 3 + N_id_2) * h->pss->corr_stride;
}
