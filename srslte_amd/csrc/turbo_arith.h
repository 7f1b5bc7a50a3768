// turbo_arith.h -- arithmetic policies and trellis helpers shared by the turbo decoder kernels (turbo_kernels.hip: throughput kernel,
// turbo_lat_kernels.hip: latency kernel).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace turbo {

typedef short s2 __attribute__((ext_vector_type(2)));

#define TD_INF 10000 // turbodecoder_win.h:56 / turbodecoder_gen.c:37
#define TD_WIN_OVERLAP 40
__device__ __forceinline__ s2 from_u(uint32_t u)
{
  return __builtin_bit_cast(s2, u);
}
__device__ __forceinline__ uint32_t to_u(s2 v)
{
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ s2 splat(short v)
{
  s2 r = {v, v};
  return r;
}
__device__ __forceinline__ s2 vmax(s2 a, s2 b)
{
  return __builtin_elementwise_max(a, b);
}
__device__ __forceinline__ s2 vmin(s2 a, s2 b)
{
  return __builtin_elementwise_min(a, b);
}

// Arithmetic of the 16-bit window decoders (WINIMP_IS_SSE16 / AVX16, turbodecoder_win.h:60-150): saturating
// int16, INF = 10000, state metrics re-based on state 0 every second step.
struct Ar16 {
  static constexpr bool kIs8 = false;
  static constexpr int  kInf = TD_INF;
  static __device__ __forceinline__ s2 add(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
  static __device__ __forceinline__ s2 add_raw(s2 a, s2 b) { return add(a, b); } // (see Ar8)
  static __device__ __forceinline__ s2 clean(s2 v) { return v; }
  static __device__ __forceinline__ s2 sub(s2 a, s2 b) { return __builtin_elementwise_sub_sat(a, b); }
  static __device__ __forceinline__ bool norm_at(uint32_t k) { return (k & 1) == 0 && k != 0; }
  // turbodecoder_win.h:480-498 (normalize_period 2; caller checks the step index)
  static __device__ __forceinline__ void normalize(s2 (&o)[8])
  {
#pragma unroll
    for (int i = 1; i < 8; i++) {
      o[i] = sub(o[i], o[0]);
    }
    o[0] = splat(0);
  }
  static __device__ __forceinline__ s2 llr(s2 m1, s2 m0) { return sub(m1, m0); }
  static __device__ __forceinline__ short tadd(short a, short b) { return (short)(a + b); } // tail trellis, plain adds
  static __device__ __forceinline__ short conv_in(int v) { return (short)v; }
  static __device__ __forceinline__ short out16(short v) { return v; }
  // extrinsic exchange (turbodecoder_iter.h:108,115): srsran_vec_sub_sss, wrapping
  static __device__ __forceinline__ s2 ex_sub(s2 a, s2 b, bool wrap) { return a - b; }
};

// Arithmetic of the 8-bit window decoders (WINIMP_IS_SSE8 / AVX8, turbodecoder_win.h:154-300): saturating int8, INF = 0, metrics
// re-based on their maximum at every step, LLR halved.
// The int8 values live in the HIGH byte of each int16 half (v << 8, low byte zero): v_pk_add/sub_i16 with clamp then saturates the
// negative side exactly where int8 does (-128 << 8 = -32768) and the positive side needs one v_pk_min_i16 with 127 << 8 -- two
// instructions per saturating add instead of add + max + min on unshifted values, and the subtraction of the running maximum (a
// result <= 0) needs none.  Wrapping int8 arithmetic is plain int16 arithmetic on the shifted values.  The 8-bit decoder is bound
// by VALU issue (its per-step normalisation and the saturation emulation), not by HBM like the 16-bit one.
struct Ar8 {
  static constexpr bool kIs8 = true;
  static constexpr int  kInf = 0;
  static __device__ __forceinline__ s2 fix_hi(s2 v) { return vmin(v, splat((short)0x7f00)); }
  static __device__ __forceinline__ s2 add(s2 a, s2 b) { return fix_hi(__builtin_elementwise_add_sat(a, b)); }
  // add_raw + clean: the positive fix-up deferred past a max.  add_raw of a CLEAN operand (low byte zero) and one that may carry the
  // low byte 0xff of an earlier positive saturation never carries into the value byte, comparisons are decided by the value byte, and
  // clean() (one AND) turns both 0x7fff and a left-over 0xff into the exact representation -- so max(add_raw ...) followed by clean()
  // equals max(add ...).  Never two unclean operands: the branch metrics and the stored betas are clean, the state metrics are
  // cleaned right after their max.
  static __device__ __forceinline__ s2 add_raw(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
  static __device__ __forceinline__ s2 clean(s2 v) { return from_u(to_u(v) & 0xff00ff00u); }
  static __device__ __forceinline__ s2 sub(s2 a, s2 b) { return fix_hi(__builtin_elementwise_sub_sat(a, b)); }
  static __device__ __forceinline__ bool norm_at(uint32_t k) { return k != 0; }
  static __device__ __forceinline__ void normalize(s2 (&o)[8])
  {
    s2 m = vmax(o[0], o[1]);
#pragma unroll
    for (int i = 2; i < 8; i++) {
      m = vmax(m, o[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      o[i] = __builtin_elementwise_sub_sat(o[i], m); // <= 0: only the negative bound can be hit, and that one is exact
    }
  }
  // divide_output 1: (m1 - m0) >> 1 on int8 = arithmetic shift of the shifted value, then drop the bit that fell into the low byte
  static __device__ __forceinline__ s2 llr(s2 m1, s2 m0) { return from_u(to_u(sub(m1, m0) >> 1) & 0xff00ff00u); }
  static __device__ __forceinline__ short tadd(short a, short b) // sadd(), :470-478: clamps the positive side only, the negative one wraps
  {
    int z = a + b;
    return z > 0x7f00 ? (short)0x7f00 : (short)z;
  }
  static __device__ __forceinline__ short conv_in(int v) { return (short)((unsigned)v << 8); } // convert_16_to_8, into the high byte
  static __device__ __forceinline__ short out16(short v) { return (short)(v >> 8); }
  // srsran_vec_sub_bbb: saturating, except in the ragged tail of its 32-byte vector loop where it wraps
  static __device__ __forceinline__ s2 ex_sub(s2 a, s2 b, bool wrap) { return wrap ? (s2)(a - b) : sub(a, b); }
};

// turbodecoder_win.h:500-548: start state of the last sub-block from the 3 tail steps (plain adds)
template <class AR>
__device__ __forceinline__ void tail_trellis(const short* xt, const short* yt, short (&old)[8])
{
  old[0] = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    old[i] = -AR::kInf;
  }
#pragma unroll
  for (int k = 2; k >= 0; k--) {
    short x = xt[k], y = yt[k];
    short xy = AR::tadd(x, y);
    short m_b[8], nw[8];
    m_b[0] = AR::tadd(old[4], xy);
    m_b[1] = old[4];
    m_b[2] = AR::tadd(old[5], y);
    m_b[3] = AR::tadd(old[5], x);
    m_b[4] = AR::tadd(old[6], x);
    m_b[5] = AR::tadd(old[6], y);
    m_b[6] = old[7];
    m_b[7] = AR::tadd(old[7], xy);
    nw[0] = old[0];
    nw[1] = AR::tadd(old[0], xy);
    nw[2] = AR::tadd(old[1], x);
    nw[3] = AR::tadd(old[1], y);
    nw[4] = AR::tadd(old[2], y);
    nw[5] = AR::tadd(old[2], x);
    nw[6] = AR::tadd(old[3], xy);
    nw[7] = old[3];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      old[i] = m_b[i] > nw[i] ? m_b[i] : nw[i];
    }
  }
}

} // namespace turbo
} // namespace phyhip
