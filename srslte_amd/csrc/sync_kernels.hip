// sync_kernels.hip -- PSS cross-correlation search and SSS m0/m1 detection for gfx950.
//
// Reference behaviour: lib/src/phy/sync/pss.c:446-534 (+ convolution.c:113-120, pss.c:408-437) and
// lib/src/phy/sync/find_sss.c:31-192, sss.c:128-156.
//
// The reference correlates with ONE FFT of length frame+fft (e.g. 309,248 = 2^11 * 151 points for a 10 ms
// capture at 30.72 Msps), once per N_id_2.  Here the same linear convolution is evaluated by overlap-save
// with 4096-point blocks: one workgroup transforms one input block ONCE (LDS Stockham FFT, fft_device.h),
// keeps the spectrum in LDS, and for each of the three N_id_2 hypotheses multiplies by the cached filter
// spectrum, inverse-transforms, takes |.|^2 (optionally the exponential moving average of pss.c:496-503)
// and reduces a block arg-max -- the capture is read from HBM once for all three hypotheses.
#include "fft_device.h"
#include "fft_reg.h"

#include <cstdlib>
#include <cstring>
#include "hip_common.h"
#include "sync_device.h"

namespace phyhip {
namespace sync {

using namespace fft;

typedef Plan<4096, 256, 16, 16, 16, 1> BlockPlan;

struct SegLoad {
  const float2* x;  // capture
  int           i0; // capture index of segment element 0
  int           frame;
  __device__ __forceinline__ float2 operator()(int m) const
  {
    const int i = i0 + m;
    return (i >= 0 && i < frame) ? x[i] : make_float2(0.f, 0.f);
  }
};

struct SpecStore {
  float2* spec;
  __device__ __forceinline__ void operator()(int k, float2 v) const { spec[k] = v; }
};

struct ProdLoad {
  const float2* spec;
  const float2* filt; // DFT of the zero-padded replica, already scaled by 1/4096
  __device__ __forceinline__ float2 operator()(int k) const { return cmul(spec[k], filt[k]); }
};

struct PowerStore {
  float* corr;  // |conv|^2 (or its moving average) of this hypothesis, index = convolution output index
  int    off;   // block output m corresponds to convolution index i = off + m
  int    m_lo;  // first valid m (fft_size - 1)
  int    m_hi;  // one past the last m owned by this block
  int    n_out; // number of valid convolution outputs (L - 2)
  float  alpha; // EMA weight, <= 0 or >= 1: no averaging
  float  best;
  int    besti;
  __device__ __forceinline__ void operator()(int m, float2 v)
  {
    const int i = off + m;
    if (m >= m_lo && m < m_hi && i < n_out) {
      float pw = v.x * v.x + v.y * v.y; // srsran_vec_abs_square_cf
      if (alpha > 0.0f && alpha < 1.0f) {
        pw = pw * alpha + corr[i] * (1.0f - alpha); // pss.c:497-500
      }
      corr[i] = pw;
      if (pw > best || (pw == best && i < besti)) {
        best  = pw;
        besti = i;
      }
    }
  }
};

#ifdef SRSRAN_HIP_WITH_VARIANTS // round 1's workgroup-per-block correlation kernel: a measured alternative, compiled into the variants library only
__global__ __launch_bounds__(256) void pss_block_kernel(const PssParams p)
{
  extern __shared__ float2 lds_all[];
  float2*   lds  = lds_all;                    // FFT exchange image
  float2*   spec = lds_all + lds_elems(4096);  // spectrum of the input block
  __shared__ float s_best[4];
  __shared__ int   s_besti[4];

  const int cap = blockIdx.y;
  const int blk = blockIdx.x;
  const int tid = threadIdx.x;
  const float2* x  = reinterpret_cast<const float2*>(p.in) + (size_t)cap * p.in_stride;
  const float2* tw = reinterpret_cast<const float2*>(p.twiddle);

  const int i0 = blk * p.hop; // first convolution output of this block
  SegLoad   ld{x, i0 - (p.fft_size - 1), p.frame_size};
  SpecStore ss{spec};
  transform<BlockPlan, false>(lds, tid, true, tw, ld, ss);
  __syncthreads();

  for (int h = 0; h < 3; h++) {
    if (!(p.n_id_2_mask & (1 << h))) {
      continue;
    }
    ProdLoad   pl{spec, reinterpret_cast<const float2*>(p.filt) + (size_t)h * 4096};
    float*     corr = p.corr + ((size_t)cap * 3 + h) * p.corr_stride;
    PowerStore ps{corr, i0 - (p.fft_size - 1), p.fft_size - 1, p.fft_size - 1 + p.hop, p.n_out, p.ema_alpha, -1.0f, 0x7fffffff};
    transform<BlockPlan, true>(lds, tid, true, tw, pl, ps);
    // block arg-max (first maximum wins on ties, as srsran_vec_max_fi)
    float b  = ps.best;
    int   bi = ps.besti;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ob  = __shfl_down(b, o);
      int   obi = __shfl_down(bi, o);
      if (ob > b || (ob == b && obi < bi)) {
        b  = ob;
        bi = obi;
      }
    }
    if ((tid & 63) == 0) {
      s_best[tid >> 6]  = b;
      s_besti[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; w++) {
        if (s_best[w] > b || (s_best[w] == b && s_besti[w] < bi)) {
          b  = s_best[w];
          bi = s_besti[w];
        }
      }
      const size_t o = ((size_t)cap * 3 + h) * p.n_blocks + blk;
      p.part_val[o] = b;
      p.part_idx[o] = bi;
    }
    __syncthreads();
  }
}
#endif // SRSRAN_HIP_WITH_VARIANTS

// Per (capture, hypothesis): global arg-max from the block partials, then the peak / side-lobe ratio of
// pss.c:408-437 evaluated on the correlation-power array.
__global__ __launch_bounds__(256) void pss_peak_kernel(const PssParams p, PssResult* res)
{
  __shared__ float s_v[256];
  __shared__ int   s_i[256];
  __shared__ int   s_lb, s_ub;
  const int cap = blockIdx.y, h = blockIdx.x, tid = threadIdx.x;
  PssResult* r = res + (size_t)cap * 3 + h;
  if (!(p.n_id_2_mask & (1 << h))) {
    if (tid == 0) {
      r->peak_pos   = -1;
      r->peak_value = 0.f;
      r->psr        = 0.f;
    }
    return;
  }
  const float* corr = p.corr + ((size_t)cap * 3 + h) * p.corr_stride;
  float b  = -1.0f;
  int   bi = 0x7fffffff;
  for (int k = tid; k < p.n_blocks; k += 256) {
    const size_t o  = ((size_t)cap * 3 + h) * p.n_blocks + k;
    const float  v  = p.part_val[o];
    const int    vi = p.part_idx[o];
    if (v > b || (v == b && vi < bi)) {
      b  = v;
      bi = vi;
    }
  }
  s_v[tid] = b;
  s_i[tid] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      if (s_v[tid + s] > s_v[tid] || (s_v[tid + s] == s_v[tid] && s_i[tid + s] < s_i[tid])) {
        s_v[tid] = s_v[tid + s];
        s_i[tid] = s_i[tid + s];
      }
    }
    __syncthreads();
  }
  const int   peak = s_i[0];
  const float pv   = s_v[0];
  const int   len  = p.n_out + 1; // conv_output_len = L - 1
  // the slopes of the main lobe are walked by one lane (a serial chain of dependent reads): serve it from an LDS
  // copy of the neighbourhood of the peak, global memory only beyond it
  constexpr int WIN = 1024;
  __shared__ float s_win[2 * WIN];
  const int        w0 = peak - WIN;
  __syncthreads();
  for (int i = tid; i < 2 * WIN; i += 256) {
    const int g = w0 + i;
    s_win[i]    = (g >= 0 && g <= len + 1) ? corr[g] : 0.f;
  }
  __syncthreads();
  auto at = [&](int i) { return (i >= w0 && i < w0 + 2 * WIN) ? s_win[i - w0] : corr[i]; };
  if (tid == 0) {
    int pl_ub = peak + 1;
    while (at(pl_ub + 1) <= at(pl_ub) && pl_ub < len) {
      pl_ub++;
    }
    int pl_lb;
    if (peak > 2) {
      pl_lb = peak - 1;
      while (at(pl_lb - 1) <= at(pl_lb) && pl_lb > 1) {
        pl_lb--;
      }
    } else {
      pl_lb = 0;
    }
    s_lb = pl_lb;
    s_ub = pl_ub;
  }
  __syncthreads();
  const int pl_lb = s_lb, pl_ub = s_ub;
  int       dist_right = len - 1 - pl_ub;
  dist_right           = dist_right < 0 ? 0 : dist_right;
  // side lobe = max(max(corr[pl_ub .. pl_ub+dist_right-1]) or corr[pl_ub], max(corr[0 .. pl_lb-1]) or corr[0]).
  // A maximum does not depend on the order: blocks that lie wholly inside one of the two ranges contribute their
  // partial maximum (part_val, written by the correlation kernel from the very values stored in corr), the blocks cut
  // by a range boundary are scanned.
  float     m    = fmaxf(corr[pl_ub], corr[0]);
  const int r_lo = pl_ub, r_hi = pl_ub + dist_right; // right range [r_lo, r_hi), left range [0, pl_lb)
  for (int k = tid; k < p.n_blocks; k += 256) {
    const int b_lo = k * p.part_span;
    const int b_hi = min(b_lo + p.part_span, p.n_out);
    if (b_hi <= b_lo) {
      continue;
    }
    if (b_hi <= pl_lb || (b_lo >= r_lo && b_hi <= r_hi)) {
      m = fmaxf(m, p.part_val[((size_t)cap * 3 + h) * p.n_blocks + k]);
    }
  }
  for (int k = 0; k < p.n_blocks; k++) { // at most four blocks are cut (two per range)
    const int b_lo = k * p.part_span;
    const int b_hi = min(b_lo + p.part_span, p.n_out);
    if (b_hi <= b_lo || b_hi <= pl_lb || (b_lo >= r_lo && b_hi <= r_hi)) {
      continue;
    }
    const int l_hi = min(b_hi, pl_lb); // part of the left range inside this block
    for (int i = b_lo + tid; i < l_hi; i += 256) {
      m = fmaxf(m, corr[i]);
    }
    const int rr_lo = max(b_lo, r_lo), rr_hi = min(b_hi, r_hi);
    for (int i = rr_lo + tid; i < rr_hi; i += 256) {
      m = fmaxf(m, corr[i]);
    }
  }
  __syncthreads();
  s_v[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      s_v[tid] = fmaxf(s_v[tid], s_v[tid + s]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    // NB: the reference's "max or first element" semantics: with an empty range srsran_vec_max_fi returns 0,
    // i.e. the first element of the range, which the seed of `m` above already contains only when the
    // range is empty; when it is not empty corr[pl_ub] / corr[0] belong to the range anyway.
    r->peak_pos   = peak;
    r->peak_value = pv;
    r->psr        = pv / s_v[0];
  }
}

// ------------------------------------------------------------------------------------------------ SSS
// One workgroup per (capture, hypothesis): the 62 SSS sub-carriers by direct DFT (exact twiddle table; symbol and
// table staged in LDS, the sum over n split over the four waves), normalisation, c0/c1 unmasking, the 2 x 31
// correlations and their arg-max (find_sss.c) on the first wave.  fft_size <= 2048 (sss.c:38, checked by the host).
__global__ __launch_bounds__(256) void sss_kernel(const SssParams p, const PssResult* pss, SssResult* res)
{
  extern __shared__ float2 s_stage[]; // [N] symbol, [N / 64][65] its decimated 64-point transforms
  __shared__ float2 y[2][31];
  __shared__ float  corr[2][31];
  __shared__ float  s_pow[2];
  __shared__ int    s_m0;
  const int cap = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  SssResult* r = res + (size_t)cap * 3 + h;
  const int  N = p.fft_size;
  // position of the SSS symbol: explicit, or derived from the PSS peak (sync.c:757-758, FDD)
  int pos = p.sss_pos ? p.sss_pos[(size_t)cap * 3 + h] : -1;
  if (!p.sss_pos) {
    const int peak = pss[(size_t)cap * 3 + h].peak_pos;
    pos            = (peak >= 2 * (N + p.cp_ext_len)) ? peak - 2 * (N + p.cp_len) + p.cp_len : -1;
  }
  if (!(p.n_id_2_mask & (1 << h)) || pos < 0 || pos + N > p.frame_size) {
    if (lane == 0) {
      r->available = 0;
      r->m0 = r->m1 = 0;
      r->m0_value = r->m1_value = 0.f;
      r->N_id_1 = -1;
      r->sf_idx = 0;
    }
    return;
  }
  const float2* x  = reinterpret_cast<const float2*>(p.in) + (size_t)cap * p.in_stride + pos;
  const float2* tw = reinterpret_cast<const float2*>(p.twiddle); // e^{-j 2 pi i / N}
  const float*  st = p.s_tilde;
  const float*  ct = p.c_tilde;
  const float*  zt = p.z_tilde;
  // The 62 bins around DC of the N-point transform of the symbol (find_sss.c:81-84, dft_fftw.c:310-320), N = 64 D: with n = D q + r,
  //   X[b] = sum_r W_N^(b r) Y_r[b mod 64],   Y_r[k] = sum_q x[D q + r] W_64^(k q)
  // -- D 64-point transforms (lane r of the first wave runs one on its own registers, fft_reg.h) and 62 sums of D terms, instead of 62 sums
  // of N terms.  The 62 bins are different modulo 64.
  regfft::cx* s_x = reinterpret_cast<regfft::cx*>(s_stage);
  regfft::cx* s_y = s_x + N; // [r][65]: Y_r[k] at r * 65 + k
  const int   D   = N >> 6;
  for (int n = lane; n < N; n += 256) {
    s_x[n] = reinterpret_cast<const regfft::cx*>(x)[n];
  }
  __syncthreads();
  if (lane < D) {
    regfft::cx v[64];
#pragma unroll
    for (int q = 0; q < 64; q++) {
      v[q] = s_x[D * q + lane];
    }
    regfft::fft64<false>(v);
#pragma unroll
    for (int k = 0; k < 64; k++) {
      s_y[lane * 65 + k] = v[k];
    }
  }
  __syncthreads();
  if (lane < 62) {
    // mirrored + dc-removed spectrum index N/2-31+k  ->  FFT bin
    const int bin = lane < 31 ? N - 31 + lane : 1 + (lane - 31);
    const int kb  = bin & 63;
    float     re = 0.f, im = 0.f;
    int       idx = 0;
    for (int r = 0; r < D; r++) {
      const float2     w  = tw[idx];
      const regfft::cx yv = s_y[r * 65 + kb];
      re += yv.x * w.x - yv.y * w.y;
      im += yv.x * w.y + yv.y * w.x;
      idx += bin;
      idx = idx >= N ? idx - N : idx;
    }
    float2 v = make_float2(re, im);
    if (p.ce) { // find_sss.c:73-76: divide by the channel estimate
      const float2 c = reinterpret_cast<const float2*>(p.ce)[((size_t)cap * 3 + h) * 62 + lane];
      const float  d = c.x * c.x + c.y * c.y;
      v              = make_float2((v.x * c.x + v.y * c.y) / d, (v.y * c.x - v.x * c.y) / d);
    }
    y[lane & 1][lane >> 1] = v;
  }
  __syncthreads();
  if (lane < 2) { // srsran_vec_avg_power_cf
    float acc = 0.f;
    for (int i = 0; i < 31; i++) {
      acc += y[lane][i].x * y[lane][i].x + y[lane][i].y * y[lane][i].y;
    }
    s_pow[lane] = acc / 31.0f;
  }
  __syncthreads();
  if (lane < 62) {
    const int   k = lane & 1, i = lane >> 1;
    const float rms = s_pow[k] != 0.0f ? sqrtf(s_pow[k]) : 1.0f;
    const float sc  = (float)(1.0 / rms);
    const float c   = ct[(i + h + (k ? 3 : 0)) % 31];
    y[k][i]         = make_float2(y[k][i].x * sc * c, y[k][i].y * sc * c);
  }
  __syncthreads();
  for (int k = 0; k < 2; k++) {
    if (k == 1) {
      if (lane < 31) { // y1 *= z1[m0]
        const float z = zt[(lane + (s_m0 % 8)) % 31];
        y[1][lane]    = make_float2(y[1][lane].x * z, y[1][lane].y * z);
      }
      __syncthreads();
    }
    if (lane < 31) {
      const int m   = lane;
      float     acc = 0.f;
      if (p.M == 0) { // differential (find_sss.c:118-160)
        float tr = 0.f, ti = 0.f;
        for (int j = 0; j < 30; j++) {
          const float  sd = st[(j + 1 + m) % 31] * st[(j + m) % 31];
          const float2 a = y[k][j + 1], b = y[k][j];
          tr += (a.x * b.x + a.y * b.y) * sd;
          ti += (a.y * b.x - a.x * b.y) * sd;
        }
        acc = tr * tr + ti * ti;
      } else { // partial correlation with M segments (find_sss.c:42-63)
        const int Nm = 31 / p.M;
        for (int seg = 0; seg < p.M; seg++) {
          float tr = 0.f, ti = 0.f;
          for (int j = 0; j < Nm; j++) {
            const float s = st[(seg * Nm + j + m) % 31];
            tr += y[k][seg * Nm + j].x * s;
            ti += y[k][seg * Nm + j].y * s;
          }
          acc += tr * tr + ti * ti;
        }
      }
      corr[k][m] = acc;
    }
    __syncthreads();
    if (lane == 0) {
      int   bm = 0;
      float bv = corr[k][0];
      for (int m = 1; m < 31; m++) {
        if (corr[k][m] > bv) {
          bv = corr[k][m];
          bm = m;
        }
      }
      if (k == 0) {
        s_m0        = bm;
        r->m0       = bm;
        r->m0_value = bv;
      } else {
        r->m1       = bm;
        r->m1_value = bv;
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    const int m0 = (int)r->m0, m1 = (int)r->m1;
    r->available = 1;
    r->sf_idx    = m1 > m0 ? 0 : 5; // sss.c:128-135
    // sss.c:139-156 with the table of gen_sss.c:55-73 evaluated on the fly
    int a = m0, b = m1;
    if (!(b > a)) {
      const int t = a;
      a           = b;
      b           = t;
    }
    int found = -1;
    if ((r->m0_value + r->m1_value) > p.threshold && a < 30 && b >= 1 && b - 1 < 30) {
      found = 0; // unused (m0, m1) pairs read 0 from the reference's zero-initialised table
      for (int id = 0; id < 168; id++) {
        const int qp = id / 30;
        const int q  = (id + qp * (qp + 1) / 2) / 30;
        const int mp = id + q * (q + 1) / 2;
        const int t0 = mp % 31, t1 = (t0 + mp / 31 + 1) % 31;
        if (t0 == a && t1 == b) {
          found = id;
        }
      }
    }
    r->N_id_1 = found;
  }
}

__global__ void pack_cells_kernel(const PssResult* a, const SssResult* b, CellResult* out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out[i].peak_pos      = a[i].peak_pos;
    out[i].peak_value    = a[i].peak_value;
    out[i].psr           = a[i].psr;
    out[i].sss_available = b[i].available;
    out[i].m0            = b[i].m0;
    out[i].m1            = b[i].m1;
    out[i].m0_value      = b[i].m0_value;
    out[i].m1_value      = b[i].m1_value;
    out[i].N_id_1        = b[i].N_id_1;
    out[i].sf_idx        = b[i].sf_idx;
  }
}

hipError_t launch_pack(const PssResult* a, const SssResult* b, CellResult* out, int n, hipStream_t stream)
{
  hipLaunchKernelGGL(pack_cells_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, a, b, out, n);
  return hipGetLastError();
}

hipError_t launch_pss(const PssParams& p, PssResult* d_res, hipStream_t stream)
{
#ifdef SRSRAN_HIP_WITH_VARIANTS
  // development knob (tests/test_gpu_variants.py, in the variants library): "block" = the workgroup-per-block kernel of round 1
  if (knob(KNOB_PSS_VARIANT) == 2) {
    static bool  attr_set = false;
    const size_t lds      = (lds_elems(4096) + 4096) * sizeof(float2);
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pss_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        return e;
      }
      attr_set = true;
    }
    hipLaunchKernelGGL(pss_block_kernel, dim3(p.n_blocks, p.n_cap), dim3(256), lds, stream, p);
  } else
#endif
  {
    hipError_t e = launch_pss_wave_blocks(p, stream);
    if (e != hipSuccess) {
      return e;
    }
  }
  hipLaunchKernelGGL(pss_peak_kernel, dim3(3, p.n_cap), dim3(256), 0, stream, p, d_res);
  return hipGetLastError();
}

hipError_t launch_sss(const SssParams& p, const PssResult* d_pss, SssResult* d_res, hipStream_t stream)
{
  hipLaunchKernelGGL(sss_kernel, dim3(3, p.n_cap), dim3(256), ((size_t)p.fft_size + 65 * (size_t)(p.fft_size / 64)) * sizeof(float2), stream, p, d_pss, d_res);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Small kernels behind the srsran_sync_t glue (sync.c), srsran_cfo_t, srsran_cp_synch_t and the PSS helpers.
// They work on one frame; none of them is a throughput path (the batched cell search above is).

// srsran_vec_prod_ccc / srsran_vec_prod_conj_ccc (a * conj(b))
__global__ void cmul_kernel(const float2* a, const float2* b, float2* out, int n, int conj_b)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float2 y = b[i];
    if (conj_b) {
      y.y = -y.y;
    }
    out[i] = cmul(a[i], y);
  }
}

hipError_t launch_cmul(const void* a, const void* b, void* out, int n, bool conj_b, hipStream_t stream)
{
  hipLaunchKernelGGL(cmul_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float2*)a, (const float2*)b, (float2*)out, n,
                     conj_b ? 1 : 0);
  return hipGetLastError();
}

// out = sa * a + sb * b (real weights): srsran_vec_sc_prod_cfc + srsran_vec_sub_ccc of pss.c:552-553
__global__ void lincomb_kernel(const float2* a, float sa, const float2* b, float sb, float2* out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float2 x = a[i], y = b[i];
    out[i] = make_float2(sa * x.x + sb * y.x, sa * x.y + sb * y.y);
  }
}

hipError_t launch_lincomb(const void* a, float sa, const void* b, float sb, void* out, int n, hipStream_t stream)
{
  hipLaunchKernelGGL(lincomb_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float2*)a, sa, (const float2*)b, sb,
                     (float2*)out, n);
  return hipGetLastError();
}

// srsran_vec_dot_prod_ccc / _conj_ccc / avg_power_cf: one workgroup per job
__global__ __launch_bounds__(256) void dot_kernel(const DotJobs jobs, float2* out)
{
  __shared__ float2 red[256];
  const int     j = blockIdx.x, tid = threadIdx.x;
  const float2* a = reinterpret_cast<const float2*>(jobs.a[j]);
  const float2* b = reinterpret_cast<const float2*>(jobs.b[j]);
  const int     n = jobs.n[j], mode = jobs.mode[j];
  float2        acc = make_float2(0.f, 0.f);
  for (int i = tid; i < n; i += 256) {
    const float2 x = a[i];
    if (mode == DOT_POWER) {
      acc.x += x.x * x.x + x.y * x.y;
    } else {
      float2 y = b[i];
      if (mode == DOT_CONJ) {
        y.y = -y.y;
      }
      const float2 pr = cmul(x, y);
      acc.x += pr.x;
      acc.y += pr.y;
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[tid].x += red[tid + s].x;
      red[tid].y += red[tid + s].y;
    }
    __syncthreads();
  }
  if (tid == 0) {
    float2 r = red[0];
    if (mode == DOT_POWER) {
      r.x /= (float)n;
    }
    out[j] = r;
  }
}

hipError_t launch_dots(const DotJobs& jobs, void* d_out, hipStream_t stream)
{
  hipLaunchKernelGGL(dot_kernel, dim3(jobs.count), dim3(256), 0, stream, jobs, (float2*)d_out);
  return hipGetLastError();
}

// srsran_cp_synch (cp.c:60-79): corr[i] = mean over the symbols of sum_k x[i+k] conj(x[i+k+N]); arg-max of |corr|
__global__ __launch_bounds__(256) void cp_synch_kernel(const float2* in, float2* corr, int max_offset, int nof_symbols, int cp_len, int N,
                                                       int* argmax)
{
  __shared__ float s_v[256];
  __shared__ int   s_i[256];
  const int tid = threadIdx.x;
  float best = -1.f;
  int   besti = 0x7fffffff;
  for (int i = tid; i < max_offset; i += 256) {
    float2        acc = make_float2(0.f, 0.f);
    const float2* ptr = in;
    for (int n = 0; n < nof_symbols; n++) {
      const int cplen = (n % 7) ? cp_len : cp_len + 1;
      float2    d = make_float2(0.f, 0.f);
      for (int k = 0; k < cplen; k++) {
        const float2 x = ptr[i + k], y = ptr[i + k + N];
        d.x += x.x * y.x + x.y * y.y;
        d.y += x.y * y.x - x.x * y.y;
      }
      acc.x += d.x / (float)nof_symbols;
      acc.y += d.y / (float)nof_symbols;
      ptr += N + cplen;
    }
    corr[i] = acc;
    const float m = acc.x * acc.x + acc.y * acc.y;
    if (m > best) {
      best  = m;
      besti = i;
    }
  }
  s_v[tid] = best;
  s_i[tid] = besti;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      if (s_v[tid + s] > s_v[tid] || (s_v[tid + s] == s_v[tid] && s_i[tid + s] < s_i[tid])) {
        s_v[tid] = s_v[tid + s];
        s_i[tid] = s_i[tid + s];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    *argmax = max_offset > 0 ? s_i[0] : 0;
  }
}

hipError_t launch_cp_synch(const void* in, void* corr, int max_offset, int nof_symbols, int cp_len, int N, int* d_argmax, hipStream_t stream)
{
  hipLaunchKernelGGL(cp_synch_kernel, dim3(1), dim3(256), 0, stream, (const float2*)in, (float2*)corr, max_offset, nof_symbols, cp_len,
                     N, d_argmax);
  return hipGetLastError();
}

// srsran_filt_decim_cc_execute (filter.c:95-110): keep every M-th sample behind num_taps-1 zeros, then the FIR
__global__ void decim_kernel(const float2* in, float2* out, int n_out, int M, int num_taps, float t0, float t1, float t2, float t3)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) {
    return;
  }
  const float taps[4] = {t0, t1, t2, t3};
  float2      acc     = make_float2(0.f, 0.f);
  for (int t = 0; t < num_taps; t++) {
    const int j = i + t - (num_taps - 1); // index into the down-sampled sequence
    if (j >= 0) {
      const float2 x = in[(size_t)j * M];
      acc.x += x.x * taps[t];
      acc.y += x.y * taps[t];
    }
  }
  out[i] = acc;
}

hipError_t launch_decim(const void* in, void* out, int n_out, int M, const float* taps4, hipStream_t stream)
{
  hipLaunchKernelGGL(decim_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, (const float2*)in, (float2*)out, n_out, M, 4,
                     taps4[0], taps4[1], taps4[2], taps4[3]);
  return hipGetLastError();
}

// srsran_pss_find_pss with frame_size < fft_size (pss.c:476-481): conv[i] = sum_n replica[n] x[i+n], then the same
// |.|^2 / moving average / arg-max as the FFT path.  One thread per output, partial maxima per workgroup.
__global__ __launch_bounds__(256) void pss_direct_kernel(const PssParams p, const float2* replica)
{
  __shared__ float s_v[256];
  __shared__ int   s_i[256];
  const int     tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  const float2* x   = reinterpret_cast<const float2*>(p.in);
  float         pw  = -1.f;
  if (i < p.n_out) {
    float2 acc = make_float2(0.f, 0.f);
    for (int n = 0; n < p.fft_size; n++) {
      const float2 pr = cmul(replica[n], x[i + n]);
      acc.x += pr.x;
      acc.y += pr.y;
    }
    pw = acc.x * acc.x + acc.y * acc.y;
    if (p.ema_alpha > 0.0f && p.ema_alpha < 1.0f) {
      pw = pw * p.ema_alpha + p.corr[i] * (1.0f - p.ema_alpha);
    }
    p.corr[i] = pw;
  }
  s_v[tid] = pw;
  s_i[tid] = i;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      if (s_v[tid + s] > s_v[tid] || (s_v[tid + s] == s_v[tid] && s_i[tid + s] < s_i[tid])) {
        s_v[tid] = s_v[tid + s];
        s_i[tid] = s_i[tid + s];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    p.part_val[blockIdx.x] = s_v[0];
    p.part_idx[blockIdx.x] = s_i[0];
  }
}

// p.n_blocks = ceil(n_out / 256); hypothesis slot 0 only (n_id_2_mask = 1)
hipError_t launch_pss_direct(const PssParams& p, const void* d_replica, PssResult* d_res, hipStream_t stream)
{
  hipLaunchKernelGGL(pss_direct_kernel, dim3(p.n_blocks), dim3(256), 0, stream, p, (const float2*)d_replica);
  hipLaunchKernelGGL(pss_peak_kernel, dim3(3, 1), dim3(256), 0, stream, p, d_res);
  return hipGetLastError();
}

// srsran_vec_apply_cfo (vector_simd.c:1692-1739, AVX2 + FMA build): z[i] = x[i] e^{j 2 pi cfo i} from EIGHT recursive
// oscillators (lane k serves samples 8m + k, advancing by e^{j 2 pi cfo 8}) and a scalar oscillator for the last
// len % 8 samples.  The float recursion IS the reference's result (its phase drifts from the ideal exponential), so
// it is replayed with the same fused operations; the seeds come from the host (cexpf).
__global__ __launch_bounds__(64) void apply_cfo_kernel(const float2* x, float2* z, int len, const CfoSeeds s)
{
  const int k = threadIdx.x;
  const int nvec = len >= 8 ? (len / 8) * 8 : 0;
  if (k < 8) {
    float2       ph  = make_float2(s.phase_re[k], s.phase_im[k]);
    const float2 osc = make_float2(s.osc8_re, s.osc8_im);
    for (int i = k; i < nvec; i += 8) {
      const float2 a = x[i];
      z[i] = make_float2(__fmaf_rn(a.x, ph.x, -__fmul_rn(a.y, ph.y)), __fmaf_rn(a.x, ph.y, __fmul_rn(a.y, ph.x)));
      ph   = make_float2(__fmaf_rn(ph.x, osc.x, -__fmul_rn(ph.y, osc.y)), __fmaf_rn(ph.x, osc.y, __fmul_rn(ph.y, osc.x)));
    }
  } else if (k == 8) {
    float2       ph  = make_float2(s.tail_re, s.tail_im);
    const float2 osc = make_float2(s.osc1_re, s.osc1_im);
    for (int i = nvec; i < len; i++) {
      const float2 a = x[i];
      z[i] = make_float2(__fsub_rn(__fmul_rn(a.x, ph.x), __fmul_rn(a.y, ph.y)), __fadd_rn(__fmul_rn(a.x, ph.y), __fmul_rn(a.y, ph.x)));
      ph   = make_float2(__fsub_rn(__fmul_rn(ph.x, osc.x), __fmul_rn(ph.y, osc.y)), __fadd_rn(__fmul_rn(ph.x, osc.y), __fmul_rn(ph.y, osc.x)));
    }
  }
}

hipError_t launch_apply_cfo(const void* x, void* z, int len, const CfoSeeds& seeds, hipStream_t stream)
{
  hipLaunchKernelGGL(apply_cfo_kernel, dim3(1), dim3(64), 0, stream, (const float2*)x, (float2*)z, len, seeds);
  return hipGetLastError();
}

} // namespace sync
} // namespace phyhip
