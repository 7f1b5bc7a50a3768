// dft_kernels.hip -- generic batched 1-D complex DFT of any length N <= 4096 for gfx950
// (srsran_dft_run_c / srsran_dft_run_guru_c / srsran_dft_precoding of the reference).
//
//   * N = 2^a 3^b 5^c : Stockham passes of radix 16/8/4/2/3/5 between two LDS images (one barrier per
//     pass), one workgroup per transform, strided/batched in and out ("guru" plans of dft_fftw.c:170-206)
//   * any other N      : direct O(N^2) evaluation from the LDS image (exact twiddle table); these
//     lengths do not occur on the hot path (PSS/SSS use 128..2048, transform precoding 12*2^a3^b5^c)
// The mirror / dc / norm / dB options of srsran_dft_run_c (dft_fftw.c:297-354) are fused into the
// load and the store.
#include "dft_device.h"
#include "fft_device.h"
#include "hip_common.h"

namespace phyhip {
namespace dft {

using namespace fft;

// source index of element n of the transform input (copy_pre, dft_fftw.c:297-308); -1 = zero
__device__ __forceinline__ int pre_index(const Params& p, int n)
{
  if (p.mirror && p.backward) {
    const int len = p.N, hlen = len / 2, offset = p.dc ? 1 : 0;
    if (n < offset) {
      return -1;
    }
    if (n < len - hlen) {
      return hlen + n - offset;
    }
    return n - (len - hlen);
  }
  return n;
}

// transform output index feeding element j of the caller's output (copy_post, :310-320); -1 = untouched
__device__ __forceinline__ int post_index(const Params& p, int j)
{
  if (p.mirror && !p.backward) {
    const int len = p.N, hlen = (len + 1) / 2, offset = p.dc ? 1 : 0;
    if (j < len - hlen) {
      return hlen + j;
    }
    if (j < len - offset) {
      return offset + j - (len - hlen);
    }
    return -1;
  }
  return j;
}

template <int R, bool INV>
__device__ __forceinline__ void lds_pass(const float2* a, float2* b, int N, int NS, const float2* __restrict__ tw)
{
  const int NB = N / R;
  for (int q = threadIdx.x; q < NB; q += blockDim.x) {
    float2    v[R];
    const int k = q % NS;
#pragma unroll
    for (int r = 0; r < R; r++) {
      v[r] = a[q + r * NB];
    }
    if (NS > 1) {
      float2 w[R];
      float2 w1 = tw[k * (N / (NS * R))];
      if (INV) {
        w1.y = -w1.y;
      }
      w[1] = w1;
#pragma unroll
      for (int r = 2; r < R; r++) {
        w[r] = (r & 1) ? cmul(w[r - 1], w1) : cmul(w[r / 2], w[r / 2]);
      }
#pragma unroll
      for (int r = 1; r < R; r++) {
        v[r] = cmul(v[r], w[r]);
      }
    }
    Dft<R, INV>::run(v);
    const int j0 = (q / NS) * NS * R + k;
#pragma unroll
    for (int r = 0; r < R; r++) {
      b[j0 + r * NS] = v[r];
    }
  }
}

template <bool INV>
__device__ void run_passes(float2*& a, float2*& b, const Params& p, const float2* tw)
{
  int NS = 1;
  for (int i = 0; i < p.npass; i++) {
    const int R = p.radix[i];
    switch (R) {
      case 16:
        lds_pass<16, INV>(a, b, p.N, NS, tw);
        break;
      case 8:
        lds_pass<8, INV>(a, b, p.N, NS, tw);
        break;
      case 5:
        lds_pass<5, INV>(a, b, p.N, NS, tw);
        break;
      case 4:
        lds_pass<4, INV>(a, b, p.N, NS, tw);
        break;
      case 3:
        lds_pass<3, INV>(a, b, p.N, NS, tw);
        break;
      default:
        lds_pass<2, INV>(a, b, p.N, NS, tw);
        break;
    }
    NS *= R;
    __syncthreads();
    float2* t = a;
    a         = b;
    b         = t;
  }
}

__global__ __launch_bounds__(256) void dft_kernel(const Params p)
{
  extern __shared__ float2 lds[];
  const int     N  = p.N;
  float2*       a  = lds;
  float2*       b  = lds + N;
  const float2* tw = reinterpret_cast<const float2*>(p.twiddle);
  const float2* in = reinterpret_cast<const float2*>(p.in) + (long)blockIdx.x * p.idist;
  float2*       out = reinterpret_cast<float2*>(p.out) + (long)blockIdx.x * p.odist;

  // real transforms (srsran_dft_run_r, dft_fftw.c:365-384; FFTW's r2r half-complex layout: hc[k] = Re X[k] for
  // k <= N/2, hc[N-k] = Im X[k] for 0 < k < N/2): evaluated through the complex engine
  const float* rin  = reinterpret_cast<const float*>(p.in) + (long)blockIdx.x * p.idist;
  float*       rout = reinterpret_cast<float*>(p.out) + (long)blockIdx.x * p.odist;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    if (p.real_mode == 1) {
      a[n] = make_float2(rin[n], 0.f);
    } else if (p.real_mode == 2) {
      const int k = n <= N / 2 ? n : N - n; // Hermitian symmetry: X[N-k] = conj(X[k])
      float     re = rin[k], im = (k == 0 || 2 * k == N) ? 0.f : rin[N - k];
      a[n]         = make_float2(re, n <= N / 2 ? im : -im);
    } else {
      const int s = pre_index(p, n);
      a[n]        = s < 0 ? make_float2(0.f, 0.f) : in[(long)s * p.istride];
    }
  }
  __syncthreads();
  if (p.npass > 0) {
    if (p.backward) {
      run_passes<true>(a, b, p, tw);
    } else {
      run_passes<false>(a, b, p, tw);
    }
  } else {
    // direct evaluation: X[k] = sum_n x[n] w^(n k mod N)
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
      float2 acc = make_float2(0.f, 0.f);
      int    idx = 0;
      for (int n = 0; n < N; n++) {
        float2 w = tw[idx];
        if (p.backward) {
          w.y = -w.y;
        }
        const float2 x = a[n];
        acc.x += x.x * w.x - x.y * w.y;
        acc.y += x.x * w.y + x.y * w.x;
        idx += k;
        idx = idx >= N ? idx - N : idx;
      }
      b[k] = acc;
    }
    __syncthreads();
    float2* t = a;
    a         = b;
    b         = t;
  }
  if (p.real_mode) {
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      float v;
      if (p.real_mode == 1) {
        v = j <= N / 2 ? a[j].x : a[N - j].y;
      } else {
        v = a[j].x;
      }
      if (p.norm != 0.0f) {
        v *= p.norm;
      }
      if (p.db) {
        v = 10.0f * log10f(v); // srsran_convert_power_to_dB
      }
      rout[j] = v;
    }
    return;
  }
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const int s = post_index(p, j);
    if (s >= 0) {
      float2 v = a[s];
      if (p.norm != 0.0f) {
        v = cscale(v, p.norm);
      }
      if (p.db) {
        v = make_float2(10.0f * log10f(v.x), 0.f); // dft_fftw.c:348-351: complex -> float takes the real part
      }
      out[(long)j * p.ostride] = v;
    }
  }
}

// ---- N > 4096: four-step decomposition N = N1 * N2 driven from the host (dft_host.cpp) with the kernel above for
// the two sets of short transforms; these two kernels supply the twiddle step and the mirror / dc re-ordering
// of srsran_dft_run_c on the full length.

// y[k1 * N2 + n2] *= e^{-+ j 2 pi k1 n2 / N}
__global__ void large_twiddle_kernel(float2* y, int N1, int N2, int backward)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long N = (long)N1 * N2;
  if (i >= N) {
    return;
  }
  const long k1 = i / N2, n2 = i - k1 * N2;
  double     sn, cs;
  sincospi(-2.0 * (double)((k1 * n2) % N) / (double)N, &sn, &cs);
  if (backward) {
    sn = -sn;
  }
  const float2 v = y[i];
  y[i] = make_float2((float)(v.x * cs - v.y * sn), (float)(v.x * sn + v.y * cs));
}

// copy_pre (backward) or copy_post (forward) of dft_fftw.c:297-320 on N elements
__global__ void large_reorder_kernel(const float2* in, float2* out, int N, int backward, int dc)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) {
    return;
  }
  Params q;
  q.N        = N;
  q.mirror   = 1;
  q.dc       = dc;
  q.backward = backward;
  const int s = backward ? pre_index(q, j) : post_index(q, j);
  if (backward) {
    out[j] = s < 0 ? make_float2(0.f, 0.f) : in[s];
  } else if (s >= 0) {
    out[j] = in[s];
  }
}

hipError_t launch_large_twiddle(void* y, int N1, int N2, bool backward, hipStream_t stream)
{
  const long N = (long)N1 * N2;
  hipLaunchKernelGGL(large_twiddle_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, (float2*)y, N1, N2, backward ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_large_reorder(const void* in, void* out, int N, bool backward, bool dc, hipStream_t stream)
{
  hipLaunchKernelGGL(large_reorder_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const float2*)in, (float2*)out, N,
                     backward ? 1 : 0, dc ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch(const Params& p, hipStream_t stream)
{
  if (has_fixed_plan(p)) {
    return launch_fixed(p, stream);
  }
  hipLaunchKernelGGL(dft_kernel, dim3(p.how_many), dim3(256), 2 * (size_t)p.N * sizeof(float2), stream, p);
  return hipGetLastError();
}

} // namespace dft
} // namespace phyhip
