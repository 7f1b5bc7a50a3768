// fft_reg.h -- small transforms on a lane's OWN registers (no LDS, no other lane): building blocks of pss_wave_kernels.hip (4096 = 64 x 64)
// and of the SSS symbol transform (sync_kernels.hip).  Every index and twiddle is a compile-time constant after unrolling.
// Sources including this are compiled without the SLP vectoriser (srslte_amd/build.py): see the note on `cx`.
#pragma once
#include <hip/hip_runtime.h>

namespace phyhip {
namespace regfft {

// Complex numbers are pairs of plain f32 registers and the arithmetic is v_add / v_mul / v_fma_f32: on gfx950 the packed-f32 forms
// (v_pk_add/mul/fma_f32) have the same peak (64 flop per clock and SIMD) and cost a lone wave ~20 cycles of issue each -- measured here:
// the packed version of this kernel took 23 cycles per vector instruction, 2.9 ms per 256 captures.
struct cx {
  float x, y;
};
static __device__ __forceinline__ cx operator+(cx a, cx b) { return {a.x + b.x, a.y + b.y}; }
static __device__ __forceinline__ cx operator-(cx a, cx b) { return {a.x - b.x, a.y - b.y}; }
// a + w b and a - w b for w = -i (forward transform) or +i (inverse): the rotation is a renaming of registers
template <bool INV>
static __device__ __forceinline__ cx add_rot(cx a, cx b)
{
  return INV ? cx{a.x - b.y, a.y + b.x} : cx{a.x + b.y, a.y - b.x};
}
template <bool INV>
static __device__ __forceinline__ cx sub_rot(cx a, cx b)
{
  return INV ? cx{a.x + b.y, a.y - b.x} : cx{a.x - b.y, a.y + b.x};
}
// z * w (CONJ: z * conj(w)): two multiplications, two fused multiply-adds
template <bool CONJ>
static __device__ __forceinline__ cx cmul2(cx z, cx w)
{
  const float wy = CONJ ? -w.y : w.y;
  return {__builtin_fmaf(-z.y, wy, z.x * w.x), __builtin_fmaf(z.x, wy, z.y * w.x)};
}

// 8-point transform in place, natural order in and out
template <bool INV>
static __device__ __forceinline__ void fft8(cx (&x)[8])
{
  constexpr float c = 0.70710678118654752440f;
  const cx a0 = x[0] + x[4], a1 = x[0] - x[4], a2 = x[2] + x[6], a3 = x[2] - x[6];
  const cx a4 = x[1] + x[5], a5 = x[1] - x[5], a6 = x[3] + x[7], a7 = x[3] - x[7];
  const cx b0 = a0 + a2, b2 = a0 - a2, b1 = add_rot<INV>(a1, a3), b3 = sub_rot<INV>(a1, a3);
  const cx b4 = a4 + a6, b6 = a4 - a6, b5 = add_rot<INV>(a5, a7), b7 = sub_rot<INV>(a5, a7);
  // W8 b5 = c (b5 + rot b5), W8^3 b7 = -c (b7 - rot b7) with rot = -i (forward) / +i (inverse)
  const cx r5 = add_rot<INV>(b5, b5), r7 = sub_rot<INV>(b7, b7);
  const cx t5 = {c * r5.x, c * r5.y}, t7 = {-c * r7.x, -c * r7.y};
  x[0] = b0 + b4;
  x[4] = b0 - b4;
  x[2] = add_rot<INV>(b2, b6);
  x[6] = sub_rot<INV>(b2, b6);
  x[1] = b1 + t5;
  x[5] = b1 - t5;
  x[3] = b3 + t7;
  x[7] = b3 - t7;
}

// e^{-+ 2 pi i J / 64} (forward: minus) as a compile-time constant
template <int J, bool INV>
static __device__ __forceinline__ cx w64()
{
  // cos(2 pi j / 64), j = 0 ... 16
  constexpr float C[17] = {1.0f,
                           0.99518472667219688624f,
                           0.98078528040323044913f,
                           0.95694033573220886494f,
                           0.92387953251128675613f,
                           0.88192126434835502971f,
                           0.83146961230254523708f,
                           0.77301045336273696081f,
                           0.70710678118654752440f,
                           0.63439328416364549822f,
                           0.55557023301960222474f,
                           0.47139673682599764856f,
                           0.38268343236508977173f,
                           0.29028467725446236764f,
                           0.19509032201612826785f,
                           0.09801714032956060199f,
                           0.0f};
  constexpr int   q  = (J & 63) >> 4, r = J & 15;
  constexpr float co = q == 0 ? C[r] : (q == 1 ? -C[16 - r] : (q == 2 ? -C[r] : C[16 - r]));
  constexpr float si = q == 0 ? C[16 - r] : (q == 1 ? C[r] : (q == 2 ? -C[16 - r] : -C[r]));
  return {co, INV ? si : -si};
}

template <int N2, int K1, bool INV>
static __device__ __forceinline__ cx inner_twiddle(cx z)
{
  constexpr int j = (N2 * K1) & 63;
  if constexpr (j == 0) {
    return z;
  } else if constexpr (j == 16) {
    return add_rot<INV>(cx{0.f, 0.f}, z);
  } else {
    return cmul2<false>(z, w64<j, INV>());
  }
}

template <int N2, bool INV>
static __device__ __forceinline__ void fft64_column(cx (&a)[64])
{
  cx t[8];
#pragma unroll
  for (int n1 = 0; n1 < 8; n1++) {
    t[n1] = a[8 * n1 + N2];
  }
  fft8<INV>(t);
  a[8 * 0 + N2] = inner_twiddle<N2, 0, INV>(t[0]);
  a[8 * 1 + N2] = inner_twiddle<N2, 1, INV>(t[1]);
  a[8 * 2 + N2] = inner_twiddle<N2, 2, INV>(t[2]);
  a[8 * 3 + N2] = inner_twiddle<N2, 3, INV>(t[3]);
  a[8 * 4 + N2] = inner_twiddle<N2, 4, INV>(t[4]);
  a[8 * 5 + N2] = inner_twiddle<N2, 5, INV>(t[5]);
  a[8 * 6 + N2] = inner_twiddle<N2, 6, INV>(t[6]);
  a[8 * 7 + N2] = inner_twiddle<N2, 7, INV>(t[7]);
  __builtin_amdgcn_sched_barrier(0); // one 8-point transform at a time: 64 of them in flight is what the scheduler would like, and 512 registers are not enough for that
}

// 64-point transform of the lane's own 64 registers, natural order in and out (8 x 8)
template <bool INV>
static __device__ __forceinline__ void fft64(cx (&a)[64])
{
  fft64_column<0, INV>(a);
  fft64_column<1, INV>(a);
  fft64_column<2, INV>(a);
  fft64_column<3, INV>(a);
  fft64_column<4, INV>(a);
  fft64_column<5, INV>(a);
  fft64_column<6, INV>(a);
  fft64_column<7, INV>(a);
  cx o[64];
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    cx t[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; n2++) {
      t[n2] = a[8 * k1 + n2];
    }
    fft8<INV>(t);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
      o[k1 + 8 * k2] = t[k2];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 64; i++) {
    a[i] = o[i];
  }
}

} // namespace regfft
} // namespace phyhip
