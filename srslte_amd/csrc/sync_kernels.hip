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
  if (tid == 0) {
    int pl_ub = peak + 1;
    while (corr[pl_ub + 1] <= corr[pl_ub] && pl_ub < len) {
      pl_ub++;
    }
    int pl_lb;
    if (peak > 2) {
      pl_lb = peak - 1;
      while (corr[pl_lb - 1] <= corr[pl_lb] && pl_lb > 1) {
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
  // side lobe = max(max(corr[pl_ub .. pl_ub+dist_right-1]) or corr[pl_ub], max(corr[0 .. pl_lb-1]) or corr[0])
  float m = fmaxf(corr[pl_ub], corr[0]);
  for (int i = tid; i < dist_right; i += 256) {
    m = fmaxf(m, corr[pl_ub + i]);
  }
  for (int i = tid; i < pl_lb; i += 256) {
    m = fmaxf(m, corr[i]);
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
// One 64-lane workgroup per (capture, hypothesis): the 62 SSS sub-carriers by direct DFT (exact twiddle
// table), normalisation, c0/c1 unmasking, the 2 x 31 correlations and their arg-max (find_sss.c).
__global__ __launch_bounds__(64) void sss_kernel(const SssParams p, const PssResult* pss, SssResult* res)
{
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
  if (lane < 62) {
    // mirrored + dc-removed spectrum index N/2-31+lane  ->  FFT bin (find_sss.c:81-84, dft_fftw.c:310-320)
    const int bin = lane < 31 ? N - 31 + lane : 1 + (lane - 31);
    float     re = 0.f, im = 0.f;
    int       idx = 0;
    for (int n = 0; n < N; n++) {
      const float2 w = tw[idx];
      const float2 v = x[n];
      re += v.x * w.x - v.y * w.y;
      im += v.x * w.y + v.y * w.x;
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
  static bool attr_set = false;
  const size_t lds = (lds_elems(4096) + 4096) * sizeof(float2);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pss_block_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      return e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(pss_block_kernel, dim3(p.n_blocks, p.n_cap), dim3(256), lds, stream, p);
  hipLaunchKernelGGL(pss_peak_kernel, dim3(3, p.n_cap), dim3(256), 0, stream, p, d_res);
  return hipGetLastError();
}

hipError_t launch_sss(const SssParams& p, const PssResult* d_pss, SssResult* d_res, hipStream_t stream)
{
  hipLaunchKernelGGL(sss_kernel, dim3(3, p.n_cap), dim3(64), 0, stream, p, d_pss, d_res);
  return hipGetLastError();
}

} // namespace sync
} // namespace phyhip
