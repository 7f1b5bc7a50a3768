// fft_device.h -- LDS-staged Stockham FFT building blocks for gfx950 (device code, header only).
//
// One OFDM symbol (N complex points) is transformed by T = N/16 (or N/8) lanes in 2..4 radix passes.
// Pass 0 takes its operands straight from HBM through a loader functor (coalesced float2 loads,
// stride N/R between a lane's operands), the last pass hands its results to a storer functor
// (coalesced stores); the passes in between exchange data through one padded LDS image
// (index i -> i + i/16, which makes both the radix-strided writes and the unit-stride reads
// bank-conflict free for ds_write_b64 / ds_read_b64).
#pragma once
#include <hip/hip_runtime.h>

namespace phyhip {
namespace fft {

__device__ __forceinline__ float2 cadd(float2 a, float2 b)
{
  return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b)
{
  return make_float2(a.x - b.x, a.y - b.y);
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cscale(float2 a, float s)
{
  return make_float2(a.x * s, a.y * s);
}
// multiply by -i (forward transforms) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 mul_mi(float2 a)
{
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// multiply by the constant e^{-+ j theta} given (c, s) = (cos theta, sin theta); forward uses e^{-j theta}
template <bool INV>
__device__ __forceinline__ float2 mul_w(float2 a, float c, float s)
{
  return INV ? make_float2(a.x * c - a.y * s, a.y * c + a.x * s) : make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[2])
  {
    float2 a = v[0], b = v[1];
    v[0]     = cadd(a, b);
    v[1]     = csub(a, b);
  }
};

template <bool INV>
struct Dft<3, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[3])
  {
    const float s  = 0.86602540378443864676f;
    float2      t1 = cadd(v[1], v[2]);
    float2      t2 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
    float2      t3 = mul_mi<INV>(cscale(csub(v[1], v[2]), s));
    v[0]           = cadd(v[0], t1);
    v[1]           = cadd(t2, t3);
    v[2]           = csub(t2, t3);
  }
};

template <bool INV>
struct Dft<4, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[4])
  {
    float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    float2 t2 = cadd(v[1], v[3]), t3 = mul_mi<INV>(csub(v[1], v[3]));
    v[0]      = cadd(t0, t2);
    v[1]      = cadd(t1, t3);
    v[2]      = csub(t0, t2);
    v[3]      = csub(t1, t3);
  }
};

template <bool INV>
struct Dft<5, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[5])
  {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2      a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]);
    float2      b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    float2      p1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    float2      p2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    float2      q1 = mul_mi<INV>(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
    float2      q2 = mul_mi<INV>(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
    v[0]           = cadd(v[0], cadd(a1, a2));
    v[1]           = cadd(p1, q1);
    v[4]           = csub(p1, q1);
    v[2]           = cadd(p2, q2);
    v[3]           = csub(p2, q2);
  }
};

template <bool INV>
struct Dft<8, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[8])
  {
    const float h = 0.70710678118654752440f;
    float2      a[4] = {v[0], v[2], v[4], v[6]};
    float2      b[4] = {v[1], v[3], v[5], v[7]};
    Dft<4, INV>::run(a);
    Dft<4, INV>::run(b);
    b[1] = mul_w<INV>(b[1], h, h);
    b[2] = mul_mi<INV>(b[2]);
    b[3] = mul_w<INV>(b[3], -h, h);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[k]     = cadd(a[k], b[k]);
      v[k + 4] = csub(a[k], b[k]);
    }
  }
};

template <bool INV>
struct Dft<9, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[9])
  {
    const float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f; // 2 pi / 9
    const float c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f; // 4 pi / 9
    const float c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f; // 8 pi / 9
    float2      y[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; n2++) {
      float2 t[3] = {v[n2], v[n2 + 3], v[n2 + 6]};
      Dft<3, INV>::run(t);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        y[n2][k] = t[k];
      }
    }
    // twiddles w9^(n2*k1)
    y[1][1] = mul_w<INV>(y[1][1], c1, s1);
    y[1][2] = mul_w<INV>(y[1][2], c2, s2);
    y[2][1] = mul_w<INV>(y[2][1], c2, s2);
    y[2][2] = mul_w<INV>(y[2][2], c4, s4);
#pragma unroll
    for (int k1 = 0; k1 < 3; k1++) {
      float2 t[3] = {y[0][k1], y[1][k1], y[2][k1]};
      Dft<3, INV>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < 3; k2++) {
        v[k1 + 3 * k2] = t[k2];
      }
    }
  }
};

template <bool INV>
struct Dft<16, INV> {
  static __device__ __forceinline__ void run(float2 (&v)[16])
  {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    float2      y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
      float2 t[4] = {v[n2], v[n2 + 4], v[n2 + 8], v[n2 + 12]};
      Dft<4, INV>::run(t);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        y[n2][k] = t[k];
      }
    }
    // twiddles w16^(n2*k1)
    y[1][1] = mul_w<INV>(y[1][1], c1, s1);
    y[1][2] = mul_w<INV>(y[1][2], h, h);
    y[1][3] = mul_w<INV>(y[1][3], s1, c1);
    y[2][1] = mul_w<INV>(y[2][1], h, h);
    y[2][2] = mul_mi<INV>(y[2][2]);
    y[2][3] = mul_w<INV>(y[2][3], -h, h);
    y[3][1] = mul_w<INV>(y[3][1], s1, c1);
    y[3][2] = mul_w<INV>(y[3][2], -h, h);
    y[3][3] = mul_w<INV>(y[3][3], -c1, -s1);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
      float2 t[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
      Dft<4, INV>::run(t);
#pragma unroll
      for (int k2 = 0; k2 < 4; k2++) {
        v[k1 + 4 * k2] = t[k2];
      }
    }
  }
};

__device__ __forceinline__ int lds_pad(int i)
{
  return i + (i >> 4);
}
constexpr int lds_elems(int n)
{
  return n + (n >> 4) + 1;
}

// One Stockham pass of radix R; NS = product of the radices of the earlier passes.
//   FIRST: operands come from `ld(index)`; LAST: results go to `st(index, value)`; otherwise LDS.
//   tw: table of e^{-j 2 pi i / N}, i < N (conjugated on the fly for inverse transforms).
template <int N, int T, int R, int NS, bool INV, bool FIRST, bool LAST, class Ld, class St>
__device__ __forceinline__ void pass(float2* lds, int tid, bool active, const float2* __restrict__ tw, Ld& ld, St& st)
{
  constexpr int NB  = N / R;            // butterflies in this pass
  constexpr int BPT = (NB + T - 1) / T; // butterflies per lane
  float2        v[BPT][R];
#pragma unroll
  for (int q = 0; q < BPT; q++) {
    const int b = tid + q * T;
    if ((NB % T == 0 || b < NB) && active) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int idx = b + r * NB;
        v[q][r]       = FIRST ? ld(idx) : lds[lds_pad(idx)];
      }
    }
  }
  if (!FIRST) {
    __syncthreads(); // every lane holds its operands: the LDS image may be overwritten
  }
#pragma unroll
  for (int q = 0; q < BPT; q++) {
    const int b = tid + q * T;
    if ((NB % T == 0 || b < NB) && active) {
      const int k = b % NS;
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
          v[q][r] = cmul(v[q][r], w[r]);
        }
      }
      Dft<R, INV>::run(v[q]);
      const int j0 = (b / NS) * NS * R + k;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int idx = j0 + r * NS;
        if (LAST) {
          st(idx, v[q][r]);
        } else {
          lds[lds_pad(idx)] = v[q][r];
        }
      }
    }
  }
  if (!LAST) {
    __syncthreads();
  }
}

// Compile-time plan: N = R0*R1*R2*R3 (unused radices are 1), T lanes per transform.
template <int N_, int T_, int R0_, int R1_, int R2_, int R3_>
struct Plan {
  static constexpr int N = N_, T = T_, R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
  static_assert(R0_ * R1_ * R2_ * R3_ == N_, "radices must multiply to N");
  static constexpr int NPASS = (R1_ > 1) + (R2_ > 1) + (R3_ > 1) + 1;
};

template <class P, bool INV, class Ld, class St>
__device__ __forceinline__ void transform(float2* lds, int tid, bool active, const float2* __restrict__ tw, Ld& ld, St& st)
{
  constexpr int N = P::N, T = P::T;
  pass<N, T, P::R0, 1, INV, true, P::NPASS == 1, Ld, St>(lds, tid, active, tw, ld, st);
  if constexpr (P::R1 > 1) {
    pass<N, T, P::R1, P::R0, INV, false, P::NPASS == 2, Ld, St>(lds, tid, active, tw, ld, st);
  }
  if constexpr (P::R2 > 1) {
    pass<N, T, P::R2, P::R0 * P::R1, INV, false, P::NPASS == 3, Ld, St>(lds, tid, active, tw, ld, st);
  }
  if constexpr (P::R3 > 1) {
    pass<N, T, P::R3, P::R0 * P::R1 * P::R2, INV, false, P::NPASS == 4, Ld, St>(lds, tid, active, tw, ld, st);
  }
}

} // namespace fft
} // namespace phyhip
