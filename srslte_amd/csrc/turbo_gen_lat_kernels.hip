// turbo_gen_lat_kernels.hip -- LATENCY kernel of the scalar turbo decoder (turbodecoder_gen.c: what AUTO runs for K <= 400, i.e. the
// transport blocks of small grants -- a VoLTE frame, a control message).
//
// tdec_gen_kernel (turbo_kernels.hip) gives a code block ONE lane and keeps its backward metrics in HBM: right for tens of thousands of
// small blocks, 166 us per half iteration when the call carries one grant.  The decoder has no windows: the K + 3 backward steps and the K
// forward steps of a half iteration are one dependent chain each, so what a small batch can use is the parallelism across the 8 trellis
// states.  Here a code block owns 8 lanes (lane = 8 x block + state slot, 8 blocks per wave) and runs the in-place butterflies of the
// window decoders' latency kernel (turbo_lat_common.h: partner metrics through DPP, the slot labelling rotating with the step index) on the
// scalar decoder's arithmetic: wrapping int16, INF = 10000, re-basing on state 0 every fourth step (turbodecoder_gen.c:58-198).  Everything
// of a block lives in LDS for the whole launch: the seven operand / extrinsic arrays, x + a-priori and all 8 (K + 4) backward metrics (16 (K + 4) int16,
// 13 KB at K = 400); HBM sees the input once and the decided bytes once.  Same results as tdec_gen_kernel, bit for bit
// (tests/test_gpu_turbo.py runs the K <= 400 cases on both); picked for launches of up to 2048 blocks (turbo_host.cpp).
#include "hip_common.h"
#include "turbo_arith.h"
#include "turbo_device.h"
#include "turbo_lat_common.h"

#include <type_traits>

namespace phyhip {
namespace turbo {
namespace lat {

namespace {

struct ArGen { // turbodecoder_gen.c: plain (wrapping) int16 adds, no saturation anywhere
  static __device__ __forceinline__ s2 add(s2 a, s2 b) { return a + b; }
  static __device__ __forceinline__ s2 add_raw(s2 a, s2 b) { return a + b; }
  static __device__ __forceinline__ s2 clean(s2 v) { return v; }
  static __device__ __forceinline__ s2 sub(s2 a, s2 b) { return a - b; }
  static __device__ __forceinline__ s2 llr(s2 m1, s2 m0) { return m1 - m0; }
};

__device__ __forceinline__ short w16(int v)
{
  return (short)(unsigned short)(unsigned)v;
}
constexpr uint32_t kGenLatArrays = 16; // per block in LDS, K + 4 int16 each: S P0 P1 A1 A2 E1 E2 | x + app | 8 x backward metrics

} // namespace

// operands come as the zero-extended 16-bit word (the metric lives in the low half of the packed type the butterflies work on; the high half is
// never looked at and wrapping arithmetic cannot trap on it)
__device__ __forceinline__ s2 ld(const short* q)
{
  return from_u((uint32_t) * reinterpret_cast<const uint16_t*>(q));
}
__device__ __forceinline__ void st16(short* q, s2 v)
{
  *q = v.x;
}

__global__ __launch_bounds__(64) void tdec_gen_lat_kernel(const GenParams p)
{
  extern __shared__ short lds[];
  const int      lane = threadIdx.x, grp = lane >> 3, slot = lane & 7;
  const int      cb_raw = (int)blockIdx.x * 8 + grp;
  const bool     live   = cb_raw < p.n_cb;
  const int      cb     = live ? cb_raw : p.n_cb - 1; // a dead group decodes a copy of the last block and writes nothing
  const uint32_t K = p.K, L = K + 4;
  const LaneK    lk = lane_consts(slot);
  short*         B  = lds + (size_t)grp * kGenLatArrays * L;
  short *        S = B, *P0 = B + L, *P1 = B + 2 * L, *A1 = B + 3 * L, *A2 = B + 4 * L, *E1 = B + 5 * L, *E2 = B + 6 * L, *XA = B + 7 * L, *BT = B + 8 * L;
  // state between launches (srsran_tdec_new_cb / srsran_tdec_iteration): the seven arrays of the block, one behind the other
  short* st = p.ws + (size_t)cb * 15 * L;

  if (p.n_begin == 0) {
    // turbodecoder_gen.c:238-258
    const size_t       in_off = p.desc ? (size_t)p.desc[cb].in_off : (size_t)cb * p.in_stride;
    const short*       in16   = p.input + in_off;
    const signed char* in8    = reinterpret_cast<const signed char*>(p.input) + in_off;
    auto               in     = [&](uint32_t i) -> short { return p.in_is8 ? (short)in8[i] : in16[i]; };
    for (uint32_t i = slot; i < K; i += 8) {
      S[i]  = in(3 * i);
      P0[i] = in(3 * i + 1);
      P1[i] = in(3 * i + 2);
    }
    if (slot < 3) {
      const uint32_t i = K + slot;
      S[i]  = in(3 * K + 2 * slot);
      P0[i] = in(3 * K + 2 * slot + 1);
      A2[i] = in(3 * K + 6 + 2 * slot);
      P1[i] = in(3 * K + 6 + 2 * slot + 1);
    }
  } else {
    for (uint32_t i = slot; i < 7 * L; i += 8) {
      B[i] = st[i];
    }
  }
  const uint16_t* inter   = p.inter;
  const uint16_t* deinter = p.deinter;
  const uint32_t  g24     = p.crc_poly & 0xffffffu;
  uint32_t        n_run   = p.n_end;
  bool            crc_good = false;
  __syncthreads();

  for (uint32_t n = p.n_begin; n < p.n_end; n++) {
    const bool dec1    = !(n & 1);
    const bool has_app = dec1 && n > 0;
    // turbodecoder_gen.c:200-236 (the extrinsic bookkeeping in front of a half iteration); with a-priori values the systematic operand of both
    // passes is x + app (:75-78, :133-136; the three tail steps take x alone): formed once, here
    if (dec1) {
      if (n) {
        for (uint32_t i = slot; i < K; i += 8) {
          const short a1 = w16(A1[i] - E1[i]);
          A1[i] = a1;
          XA[i] = w16(S[i] + a1);
        }
        if (slot < 3) {
          XA[K + slot] = S[K + slot];
        }
      }
    } else {
      for (uint32_t i = slot; i < K; i += 8) {
        short e = E1[i];
        if (n > 1) {
          e     = w16(e - A1[i]);
          E1[i] = e;
        }
        A2[deinter[i]] = e;
      }
    }
    const short* X   = dec1 ? (has_app ? XA : S) : A2;
    const short* Y   = dec1 ? P0 : P1;
    short*       Out = dec1 ? E1 : E2;
    __syncthreads();

    // ---- map_gen_beta (:58-112): beta(K + 3) = {0, -INF ...}; beta(k) from beta(k + 1), all of them kept; re-based on state 0 at k = 0 mod 4, k < K.
    // Twelve steps at a time with their operands fetched up front: a store into the metrics may alias everything the compiler knows about, so a
    // load behind it waits for it -- one LDS round trip per STEP in the dependent chain (52 us per half iteration at K = 176 that way, 2.8x the
    // one-lane kernel instead of 10x).  Twelve = residues of the labelling (3) and of the re-basing (4) both static inside a chunk.
    s2 b = splat(slot == 0 ? (short)0 : (short)-TD_INF);
    st16(&BT[(K + 3) * 8 + slot], b);
    auto beta_one = [&](auto rtag, int k, s2 x, s2 y) {
      constexpr int R = decltype(rtag)::value;
      s2            go, gc;
      gammas<ArGen, R>(lk, x, y, go, gc);
      b = beta_step<ArGen, R>(b, go, gc);
      st16(&BT[k * 8 + slot], b);
    };
    auto rebase = [&](s2& m) { m = m - from_u(bcast_slot0(to_u(m))); };
    {
      int k = (int)K + 2;
      for (; k >= 0 && k % 12 != 11; k--) {
        const s2 x = ld(&X[k]), y = ld(&Y[k]);
        if (k % 3 == 2) {
          beta_one(std::integral_constant<int, 2>{}, k, x, y);
        } else if (k % 3 == 1) {
          beta_one(std::integral_constant<int, 1>{}, k, x, y);
        } else {
          beta_one(std::integral_constant<int, 0>{}, k, x, y);
        }
        if ((k & 3) == 0 && (uint32_t)k < K) {
          rebase(b);
        }
      }
      for (; k >= 11; k -= 12) {
        s2 x[12], y[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
          x[i] = ld(&X[k - i]);
          y[i] = ld(&Y[k - i]);
        }
        const bool body = (uint32_t)k < K; // (a chunk that reaches into the tail steps: k - i < K decides per step)
#pragma unroll
        for (int i = 0; i < 12; i += 3) { // k - i = 11, 8, 5, 2 mod 12
          beta_one(std::integral_constant<int, 2>{}, k - i, x[i], y[i]);
          if (((11 - i) & 3) == 0 && (body || (uint32_t)(k - i) < K)) {
            rebase(b);
          }
          beta_one(std::integral_constant<int, 1>{}, k - i - 1, x[i + 1], y[i + 1]);
          if (((10 - i) & 3) == 0 && (body || (uint32_t)(k - i - 1) < K)) {
            rebase(b);
          }
          beta_one(std::integral_constant<int, 0>{}, k - i - 2, x[i + 2], y[i + 2]);
          if (((9 - i) & 3) == 0 && (body || (uint32_t)(k - i - 2) < K)) {
            rebase(b);
          }
        }
      }
    }
    __syncthreads();

    // ---- map_gen_alpha (:114-198): alpha(0) = {0, -INF ...}; step j: max1 - max0 over the branches into alpha(j + 1) against beta(j + 1); re-based
    // when j + 1 = 0 mod 4.  All 8 lanes of a block hold the output: all of them store it (same value, same address).
    s2   a = splat(slot == 0 ? (short)0 : (short)-TD_INF);
    auto alpha_one = [&](auto rtag, uint32_t j, s2 x, s2 y, s2 bn) {
      constexpr int R = decltype(rtag)::value;
      s2            go, gc;
      gammas<ArGen, R>(lk, x, y, go, gc);
      st16(&Out[j], alpha_step<ArGen, R, true>(lk, a, bn, go, gc));
    };
    {
      uint32_t j = 0;
      for (; j + 12 <= K; j += 12) {
        s2 x[12], y[12], bn[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
          x[i]  = ld(&X[j + i]);
          y[i]  = ld(&Y[j + i]);
          bn[i] = ld(&BT[(j + i + 1) * 8 + slot]);
        }
#pragma unroll
        for (int i = 0; i < 12; i += 3) {
          alpha_one(std::integral_constant<int, 0>{}, j + i, x[i], y[i], bn[i]);
          if (((i + 1) & 3) == 0) {
            rebase(a);
          }
          alpha_one(std::integral_constant<int, 1>{}, j + i + 1, x[i + 1], y[i + 1], bn[i + 1]);
          if (((i + 2) & 3) == 0) {
            rebase(a);
          }
          alpha_one(std::integral_constant<int, 2>{}, j + i + 2, x[i + 2], y[i + 2], bn[i + 2]);
          if (((i + 3) & 3) == 0) {
            rebase(a);
          }
        }
      }
      for (; j < K; j++) { // (K is a multiple of 8: up to 8 steps)
        const s2 x = ld(&X[j]), y = ld(&Y[j]), bn = ld(&BT[(j + 1) * 8 + slot]);
        if (j % 3 == 0) {
          alpha_one(std::integral_constant<int, 0>{}, j, x, y, bn);
        } else if (j % 3 == 1) {
          alpha_one(std::integral_constant<int, 1>{}, j, x, y, bn);
        } else {
          alpha_one(std::integral_constant<int, 2>{}, j, x, y, bn);
        }
        if (((j + 1) & 3) == 0) {
          rebase(a);
        }
      }
    }
    __syncthreads();
    if (!dec1) {
      for (uint32_t i = slot; i < K; i += 8) {
        A1[inter[i]] = E2[i];
      }
      __syncthreads();
    }
    n_run = n + 1;
    if (p.crc_poly) {
      // decode_tb_cb (sch.c:420-454): the checksum of the K hard bits after every half iteration (crc.c:92-140: MSB first, zero start; zero = good).
      // The 8 lanes take an eighth of the bits each and shift their remainders into place with x^(bits behind) mod g (p.crc_mult8, turbo_host.cpp).
      const short*   Cs = (n_run & 1) ? E1 : A1;
      const uint32_t c8 = (K + 7) >> 3, lo = slot * c8 < K ? slot * c8 : K, hi = lo + c8 < K ? lo + c8 : K;
      uint32_t       c  = 0;
      for (uint32_t i = lo; i < hi; i++) {
        const uint32_t x = Cs[i] > 0 ? 1u : 0u;
        c = ((c << 1) & 0xffffffu) ^ ((((c >> 23) ^ x) & 1u) ? g24 : 0u);
      }
      const uint32_t m = p.crc_mult8[slot];
      uint32_t       r = 0;
#pragma unroll 4
      for (int i = 23; i >= 0; i--) {
        r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? g24 : 0u);
        r ^= ((m >> i) & 1u) ? c : 0u;
      }
      r ^= partner<0>(r);
      r ^= partner<1>(r);
      r ^= partner<2>(r);
      if (r == 0) {
        crc_good = true;
        break;
      }
    }
  }
  if (live && slot == 0) {
    if (p.noi) {
      p.noi[cb] = (int)(n_run - p.n_begin);
    }
    if (p.crc_ok) {
      p.crc_ok[cb] = crc_good ? 1 : 0;
    }
  }
  // decision (turbodecoder.c:370-378, turbodecoder_gen.c:260-277)
  const short* Dd = (n_run & 1) ? E1 : A1;
  if (live) {
    uint8_t*       out       = p.output + (p.desc ? (size_t)p.desc[cb].out_off : (size_t)cb * p.out_stride);
    const uint32_t out_bytes = p.desc ? p.desc[cb].out_bytes : K / 8;
    for (uint32_t jb = slot; jb < out_bytes; jb += 8) {
      uint32_t byte = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        byte |= (Dd[jb * 8 + t] > 0 ? 0x80u : 0u) >> t;
      }
      out[jb] = (uint8_t)byte;
    }
    if (p.dec_llr) {
      short* o16 = p.dec_llr + (size_t)cb * K;
      for (uint32_t i = slot; i < K; i += 8) {
        o16[i] = Dd[i];
      }
    }
    if (!p.desc) { // (a transport-block launch is never resumed)
      for (uint32_t i = slot; i < 7 * L; i += 8) {
        st[i] = B[i];
      }
    }
  }
}

} // namespace lat

size_t gen_lat_lds_bytes(uint32_t K)
{
  return (size_t)8 * lat::kGenLatArrays * (K + 4) * sizeof(short);
}

hipError_t launch_gen_lat(const GenParams& p, hipStream_t stream)
{
  const size_t lds = gen_lat_lds_bytes(p.K);
  if (lds > 156 * 1024 || (p.crc_poly && !p.crc_mult8)) {
    return hipErrorInvalidValue;
  }
  static bool attr_set[kMaxDevices] = {};
  const int   dev = current_device(), di = dev >= 0 && dev < kMaxDevices ? dev : 0;
  if (!attr_set[di]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lat::tdec_gen_lat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) {
      return e;
    }
    attr_set[di] = true;
  }
  hipLaunchKernelGGL(lat::tdec_gen_lat_kernel, dim3(ceil_div(p.n_cb, 8)), dim3(64), lds, stream, p);
  return hipGetLastError();
}

} // namespace turbo
} // namespace phyhip
