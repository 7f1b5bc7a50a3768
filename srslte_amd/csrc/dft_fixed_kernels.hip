// dft_fixed_kernels.hip -- the 34 transform-precoding lengths of srsran_dft_precoding (12 * N_prb, N_prb = 2^a 3^b 5^c
// <= 100, dft_precoding.c:84-96) as compile-time plans for gfx950: contiguous, batched, optional 1/sqrt(N).
//
// The run-time-planned kernel of dft_kernels.hip spends a workgroup of 256 lanes, two LDS images and integer divisions
// by run-time strides on every transform; SC-FDMA is a pure HBM stream (2 * 8 * N bytes per transform), so
// here every length gets the OFDM engine of fft_device.h instead: T lanes per transform (T ~ N/16, chosen with the
// radix order so that every pass fills its lanes), G = 256 / T transforms per workgroup, first pass fed straight from
// HBM, last pass stored straight to HBM, one padded LDS image per transform in between, all index arithmetic
// constant-folded.  Any other use of the DFT handles (strides, mirror / dc, dB, real plans, other lengths) stays on the
// generic kernel.
#include "dft_device.h"
#include "fft_device.h"
#include "hip_common.h"

namespace phyhip {
namespace dft {

using namespace fft;

// N, lanes per transform, radices in pass order (table made by minimising the idle lane slots over radix orders and T)
#define DFT_FIXED_PLANS(X)                                                                                             \
  X(12, 1, 4, 3, 1, 1)                                                                                                 \
  X(24, 1, 8, 3, 1, 1)                                                                                                 \
  X(36, 2, 9, 4, 1, 1)                                                                                                 \
  X(48, 3, 16, 3, 1, 1)                                                                                                \
  X(60, 3, 5, 4, 3, 1)                                                                                                 \
  X(72, 9, 9, 8, 1, 1)                                                                                                 \
  X(96, 6, 16, 2, 3, 1)                                                                                                \
  X(108, 6, 9, 3, 4, 1)                                                                                                \
  X(120, 8, 8, 3, 5, 1)                                                                                                \
  X(144, 9, 16, 9, 1, 1)                                                                                               \
  X(180, 12, 9, 5, 4, 1)                                                                                               \
  X(192, 12, 8, 8, 3, 1)                                                                                               \
  X(216, 14, 9, 3, 8, 1)                                                                                               \
  X(240, 16, 16, 5, 3, 1)                                                                                              \
  X(288, 18, 9, 8, 4, 1)                                                                                               \
  X(300, 20, 5, 5, 3, 4)                                                                                               \
  X(324, 36, 9, 4, 9, 1)                                                                                               \
  X(360, 24, 9, 5, 8, 1)                                                                                               \
  X(384, 24, 16, 8, 3, 1)                                                                                              \
  X(432, 27, 16, 3, 9, 1)                                                                                              \
  X(480, 32, 16, 2, 5, 3)                                                                                              \
  X(540, 36, 9, 4, 5, 3)                                                                                               \
  X(576, 36, 9, 8, 8, 1)                                                                                               \
  X(600, 40, 8, 5, 3, 5)                                                                                               \
  X(648, 41, 9, 8, 9, 1)                                                                                               \
  X(720, 48, 16, 5, 9, 1)                                                                                              \
  X(768, 48, 16, 3, 16, 1)                                                                                             \
  X(864, 62, 16, 3, 2, 9)                                                                                              \
  X(900, 60, 9, 5, 4, 5)                                                                                               \
  X(960, 64, 8, 8, 3, 5)                                                                                               \
  X(972, 61, 9, 9, 3, 4)                                                                                               \
  X(1080, 72, 9, 8, 3, 5)                                                                                              \
  X(1152, 72, 16, 9, 8, 1)                                                                                             \
  X(1200, 80, 16, 3, 5, 5)

template <class P, int G, bool INV>
__global__ __launch_bounds__(G* P::T) void dft_fixed_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                             const float2* __restrict__ tw, long idist, long odist,
                                                             int how_many, float norm)
{
  constexpr int     IMG = lds_elems(P::N);
  __shared__ float2 lds[G * IMG];
  const int         g = threadIdx.x / P::T, tid = threadIdx.x - g * P::T;
  const long        t = (long)blockIdx.x * G + g;
  const bool        active = t < how_many;
  const float2*     src    = in + (active ? t : 0) * idist;
  float2*           dst    = out + (active ? t : 0) * odist;
  auto              ld     = [&](int i) { return src[i]; };
  auto              st     = [&](int i, float2 v) { dst[i] = norm != 0.0f ? cscale(v, norm) : v; };
  transform<P, INV>(lds + g * IMG, tid, active, tw, ld, st);
}

template <class P>
static hipError_t launch_plan(const Params& p, hipStream_t stream)
{
  constexpr int G    = 256 / P::T > 0 ? 256 / P::T : 1;
  const dim3    grid((unsigned)((p.how_many + G - 1) / G)), block(G * P::T);
  const float2* in  = reinterpret_cast<const float2*>(p.in);
  float2*       out = reinterpret_cast<float2*>(p.out);
  const float2* tw  = reinterpret_cast<const float2*>(p.twiddle);
  if (p.backward) {
    hipLaunchKernelGGL((dft_fixed_kernel<P, G, true>), grid, block, 0, stream, in, out, tw, p.idist, p.odist, p.how_many, p.norm);
  } else {
    hipLaunchKernelGGL((dft_fixed_kernel<P, G, false>), grid, block, 0, stream, in, out, tw, p.idist, p.odist, p.how_many, p.norm);
  }
  return hipGetLastError();
}

bool has_fixed_plan(const Params& p)
{
  if (p.npass == 0 || p.istride != 1 || p.ostride != 1 || p.mirror || p.db || p.real_mode || p.how_many <= 0) {
    return false;
  }
  // the first pass of a transform reads all of its input before the last pass writes, so in == out is fine, but
  // partially overlapping transforms are not (the generic kernel has the same contract)
  switch (p.N) {
#define X(N, T, R0, R1, R2, R3) case N:
    DFT_FIXED_PLANS(X)
#undef X
    return true;
    default:
      return false;
  }
}

hipError_t launch_fixed(const Params& p, hipStream_t stream)
{
  switch (p.N) {
#define X(N, T, R0, R1, R2, R3)                                                                                        \
  case N:                                                                                                              \
    return launch_plan<Plan<N, T, R0, R1, R2, R3>>(p, stream);
    DFT_FIXED_PLANS(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
}

} // namespace dft
} // namespace phyhip
