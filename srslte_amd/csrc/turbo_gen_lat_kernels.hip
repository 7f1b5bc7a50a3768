// turbo_gen_lat_kernels.hip -- LATENCY kernel of the scalar turbo decoder (turbodecoder_gen.c: what AUTO runs for K <= 400, i.e. the
// transport blocks of small grants -- a VoLTE frame, a control message).
//
// tdec_gen_kernel (turbo_kernels.hip) gives a code block ONE lane and keeps its backward metrics in HBM: right for tens of thousands of
// small blocks, 130-300 us per half iteration when the call carries one grant.  The decoder has no windows: the K + 3 backward steps and the
// K forward steps of a half iteration are one dependent chain each.  What a small batch can use:
//   * the 8 trellis states side by side: a recursion owns 8 lanes and runs the in-place butterflies of the window decoders' latency kernel
//     (turbo_lat_common.h: partner metrics through DPP, the slot labelling rotating with the step index) on the scalar decoder's arithmetic --
//     wrapping int16, INF = 10000, re-basing on state 0 every fourth step (turbodecoder_gen.c:58-198);
//   * the forward and the backward recursion side by side: neither needs the other (only the outputs need both), and a butterfly step has the
//     same form in both directions -- new = max(own + g, partner + g') -- so lanes 0-7 of a block's 16 walk alpha(0 ... K) while lanes 8-15 walk
//     beta(K + 3 ... 0) in the SAME instructions.  The backward half numbers its slots with the bits permuted (by (K + 2) mod 3) so that its
//     partner at step t is the same physical lane distance as the forward half's; its operands are kept in reversed copies so that both
//     halves address "row t"; the re-basing steps (forward: before the metric is filed, backward: after) fall on different t mod 4;
//   * all metrics of both recursions are filed in LDS (row = 8 int16 in the labelling of the time index) and the K outputs max1 - max0 are
//     formed afterwards by ALL 64 lanes, a step per lane: both rows as one 16-byte read each, the 16 branch sums and the two maxima on packed
//     int16 pairs with the lane-to-state bookkeeping folded into compile-time masks (three variants: time index mod 3).
// Everything of a block lives in LDS for the whole launch (28 (K + 4) int16 = 22 KB at K = 400; four blocks per wave); HBM sees the input once
// and the decided bytes once.  Same results as tdec_gen_kernel, bit for bit (tests/test_gpu_turbo.py runs every K <= 400 case on both).
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
};

__device__ __forceinline__ short w16(int v)
{
  return (short)(unsigned short)(unsigned)v;
}
// operands come as the zero-extended 16-bit word: the metric lives in the low half of the packed type the butterflies work on, the high half
// is never looked at and wrapping arithmetic cannot trap on it
__device__ __forceinline__ s2 ld(const short* q)
{
  return from_u((uint32_t) * reinterpret_cast<const uint16_t*>(q));
}
__device__ __forceinline__ void st16(short* q, s2 v)
{
  *q = v.x;
}

// arrays of a block in LDS, K + 4 int16 each (x + app shares E2's place: E2 is dead between the end of decoder 2 and its next run)
enum { aS = 0, aP0, aP1, aA1, aA2, aE1, aE2, aSR, aP0R, aP1R, aA2R, aXAR, aAT, aBTR = aAT + 8, kArrays = aBTR + 8 };
constexpr int aXA = aE2;

// ---- the outputs of the steps j = R mod 3: branch sums into the 8 new states against beta(j + 1), packed two slots per register.
// Slot s of row j holds alpha_j[state_of(R, s)], slot s of the beta row holds beta_{j+1}[nx], nx = state_of(R + 1, s): the slot's own metric and the
// one of slot s ^ (1 << R) are the two predecessors of nx; d0: the slot's own one sits on the data-bit-0 branch; ty: nx's branches carry {y, x}, not {0, x + y}.
__host__ __device__ constexpr bool slot_d0(int R, int s)
{
  return state_of(R, s) == pm_of(state_of(R + 1, s));
}
__host__ __device__ constexpr bool slot_ty(int R, int s)
{
  return type_y(state_of(R + 1, s));
}
__host__ __device__ constexpr uint32_t pair_mask(bool lo, bool hi)
{
  return (lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u);
}
__device__ __forceinline__ uint32_t sel(uint32_t mask, uint32_t a, uint32_t b) // mask ? a : b, bit-wise
{
  return (a & mask) | (b & ~mask);
}
template <int R>
__device__ __forceinline__ short llr_of_step(const uint4 av, const uint4 bv, s2 x, s2 y)
{
  const uint32_t a[4] = {av.x, av.y, av.z, av.w}, bq[4] = {bv.x, bv.y, bv.z, bv.w};
  const uint32_t x2 = to_u(x) | (to_u(x) << 16), y2 = to_u(y) | (to_u(y) << 16), xy2 = to_u(from_u(x2) + from_u(y2));
  s2             v0[4], v1[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const uint32_t pa = R == 0 ? __builtin_amdgcn_alignbit(a[d], a[d], 16) : (R == 1 ? a[d ^ 1] : a[d ^ 2]);
    const uint32_t md = pair_mask(slot_d0(R, 2 * d), slot_d0(R, 2 * d + 1)), mt = pair_mask(slot_ty(R, 2 * d), slot_ty(R, 2 * d + 1));
    const s2       s0 = from_u(sel(md, a[d], pa)), s1 = from_u(sel(md, pa, a[d])); // data bit 0 / 1 predecessor metric
    const s2       g0 = from_u(sel(mt, y2, 0u)), g1 = from_u(sel(mt, x2, xy2));
    v0[d] = (s0 + g0) + from_u(bq[d]);
    v1[d] = (s1 + g1) + from_u(bq[d]);
  }
  s2 m0 = vmax(vmax(v0[0], v0[1]), vmax(v0[2], v0[3])), m1 = vmax(vmax(v1[0], v1[1]), vmax(v1[2], v1[3]));
  m0    = vmax(m0, from_u(__builtin_amdgcn_alignbit(to_u(m0), to_u(m0), 16)));
  m1    = vmax(m1, from_u(__builtin_amdgcn_alignbit(to_u(m1), to_u(m1), 16)));
  return (m1 - m0).x;
}

} // namespace

__global__ __launch_bounds__(64) void tdec_gen_lat_kernel(const GenParams p)
{
  extern __shared__ __attribute__((aligned(16))) short lds[];
  const int      lane = threadIdx.x, grp = lane >> 4, l16 = lane & 15, q = lane & 7;
  const bool     back = (lane & 8) != 0; // the backward half of the block's 16 lanes
  const int      cb_raw = (int)blockIdx.x * 4 + grp;
  const bool     live   = cb_raw < p.n_cb;
  const int      cb     = live ? cb_raw : p.n_cb - 1; // a dead group decodes a copy of the last block and writes nothing
  const int      n_live = p.n_cb - (int)blockIdx.x * 4 < 4 ? p.n_cb - (int)blockIdx.x * 4 : 4;
  const uint32_t K = p.K, L = K + 4;
  short*         B = lds + (size_t)grp * kArrays * L;
  auto           arr = [&](int a) { return B + (size_t)a * L; };
  short *        S = arr(aS), *P0 = arr(aP0), *P1 = arr(aP1), *A1 = arr(aA1), *A2 = arr(aA2), *E1 = arr(aE1), *E2 = arr(aE2);
  short *        SR = arr(aSR), *P0R = arr(aP0R), *P1R = arr(aP1R), *A2R = arr(aA2R), *XA = arr(aXA), *XAR = arr(aXAR);
  // state between launches (srsran_tdec_new_cb / srsran_tdec_iteration): the first seven arrays of the block, one behind the other
  short* st = p.ws + (size_t)cb * 15 * L;

  // ---- lane constants.  Backward half: logical slot = physical lane bits permuted so that the partner bit of step t, (K + 2 - t) mod 3 in
  // logical terms, sits at physical bit t mod 3: logical bit u <-> physical bit (c - u) mod 3, c = (K + 2) mod 3
  const int c3   = (int)((K + 2) % 3);
  int       slot = q;
  if (back) {
    slot = 0;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int pb = (c3 - u + 3) % 3;
      slot |= ((q >> pb) & 1) << u;
    }
  }
  LaneK lk; // indexed by the PHYSICAL residue t mod 3
  {
    const LaneK l0 = lane_consts(slot);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int rho = back ? (c3 - r + 3) % 3 : r;
      lk.ty[r]      = rho == 0 ? l0.ty[0] : (rho == 1 ? l0.ty[1] : l0.ty[2]);
      lk.d0[r]      = rho == 0 ? l0.d0[0] : (rho == 1 ? l0.d0[1] : l0.d0[2]);
      lk.q[r]       = false;
    }
    lk.odd = false;
  }

  if (p.n_begin == 0) {
    // turbodecoder_gen.c:238-258
    const size_t       in_off = p.desc ? (size_t)p.desc[cb].in_off : (size_t)cb * p.in_stride;
    const short*       in16   = p.input + in_off;
    const signed char* in8    = reinterpret_cast<const signed char*>(p.input) + in_off;
    auto               in     = [&](uint32_t i) -> short { return p.in_is8 ? (short)in8[i] : in16[i]; };
    for (uint32_t i = l16; i < K; i += 16) {
      S[i]  = in(3 * i);
      P0[i] = in(3 * i + 1);
      P1[i] = in(3 * i + 2);
    }
    if (l16 < 3) {
      const uint32_t i = K + l16;
      S[i]  = in(3 * K + 2 * l16);
      P0[i] = in(3 * K + 2 * l16 + 1);
      A2[i] = in(3 * K + 6 + 2 * l16);
      P1[i] = in(3 * K + 6 + 2 * l16 + 1);
    }
  } else {
    for (uint32_t i = l16; i < 7 * L; i += 16) {
      B[i] = st[i];
    }
  }
  __syncthreads();
  for (uint32_t i = l16; i < K + 3; i += 16) { // reversed copies for the backward half: row t = element K + 2 - t
    SR[K + 2 - i]  = S[i];
    P0R[K + 2 - i] = P0[i];
    P1R[K + 2 - i] = P1[i];
  }
  if (l16 < 3) {
    A2R[2 - l16] = A2[K + l16];
  }
  const uint16_t* inter   = p.inter;
  const uint16_t* deinter = p.deinter;
  const uint32_t  g24     = p.crc_poly & 0xffffffu;
  uint32_t        n_run = p.n_begin, n_done = 0;
  bool            done = false, crc_good = false;

  // decision (turbodecoder.c:370-378, turbodecoder_gen.c:260-277) of a block whose last half iteration was number n_last
  auto decide = [&](uint32_t n_last) {
    const short*   Dd        = (n_last & 1) ? E1 : A1;
    uint8_t*       out       = p.output + (p.desc ? (size_t)p.desc[cb].out_off : (size_t)cb * p.out_stride);
    const uint32_t out_bytes = p.desc ? p.desc[cb].out_bytes : K / 8;
    for (uint32_t jb = l16; jb < out_bytes; jb += 16) {
      uint32_t byte = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        byte |= (Dd[jb * 8 + t] > 0 ? 0x80u : 0u) >> t;
      }
      out[jb] = (uint8_t)byte;
    }
    if (p.dec_llr) {
      short* o16 = p.dec_llr + (size_t)cb * K;
      for (uint32_t i = l16; i < K; i += 16) {
        o16[i] = Dd[i];
      }
    }
  };
  __syncthreads();

  for (uint32_t n = p.n_begin; n < p.n_end; n++) {
    const bool dec1    = !(n & 1);
    const bool has_app = dec1 && n > 0;
    // turbodecoder_gen.c:200-236 (the extrinsic bookkeeping in front of a half iteration); with a-priori values the systematic operand of both
    // passes is x + app (:75-78, :133-136; the three tail steps take x alone): formed once, here
    if (dec1) {
      if (n) {
        for (uint32_t i = l16; i < K; i += 16) {
          const short a1 = w16(A1[i] - E1[i]);
          A1[i] = a1;
          const short xa = w16(S[i] + a1);
          XA[i]          = xa;
          XAR[K + 2 - i] = xa;
        }
        if (l16 < 3) {
          XA[K + l16]  = S[K + l16];
          XAR[2 - l16] = S[K + l16];
        }
      }
    } else {
      for (uint32_t i = l16; i < K; i += 16) {
        short e = E1[i];
        if (n > 1) {
          e     = w16(e - A1[i]);
          E1[i] = e;
        }
        const uint32_t d = deinter[i];
        A2[d]            = e;
        A2R[K + 2 - d]   = e;
      }
    }
    const short* Xp = back ? (dec1 ? (has_app ? XAR : SR) : A2R) : (dec1 ? (has_app ? XA : S) : A2); // this lane's operand rows
    const short* Yp = back ? (dec1 ? P0R : P1R) : (dec1 ? P0 : P1);
    short*       M  = arr(back ? aBTR : aAT) + slot; // this lane's column of the metric rows: alpha(t) / beta(K + 3 - t) in row t
    __syncthreads();

    // ---- both recursions (map_gen_beta :58-112, map_gen_alpha :114-198 without the output): step t takes alpha(t) -> alpha(t + 1) with the operands of
    // k = t, and beta(K + 3 - t) -> beta(K + 2 - t) with the operands of k = K + 2 - t; K + 3 steps (the forward half's last three rows are never read).
    // Forward: re-based when t + 1 = 0 mod 4, BEFORE the metric is filed (the next step starts from the re-based one); backward: the metric is
    // filed as it is and re-based afterwards when k = K + 2 - t = 0 mod 4 and k < K, i.e. t = 2 mod 4, t > 2 (K is a multiple of 8).
    // Twelve steps at a time with their operands fetched up front (a load behind a store into LDS waits for it, and the compiler cannot tell the
    // arrays apart): residues of the labelling (3) and of the re-basing (4) are both static inside a chunk.
    s2 m = splat(slot == 0 ? (short)0 : (short)-TD_INF);
    st16(&M[0], m);
    auto step = [&](auto rtag, s2 x, s2 y) {
      constexpr int R = decltype(rtag)::value;
      s2            go, gc;
      gammas<ArGen, R>(lk, x, y, go, gc);
      m = beta_step<ArGen, R>(m, go, gc); // (the same form in both directions)
    };
    auto rebase_if = [&](bool mine) {
      const uint32_t bc = bcast_slot0(to_u(m));
      m                 = m - from_u(mine ? bc : 0u);
    };
    {
      const uint32_t T = K + 3;
      uint32_t       t = 0;
      for (; t + 12 <= T; t += 12) {
        s2 x[12], y[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
          x[i] = ld(&Xp[t + i]);
          y[i] = ld(&Yp[t + i]);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
          if (i % 3 == 0) {
            step(std::integral_constant<int, 0>{}, x[i], y[i]);
          } else if (i % 3 == 1) {
            step(std::integral_constant<int, 1>{}, x[i], y[i]);
          } else {
            step(std::integral_constant<int, 2>{}, x[i], y[i]);
          }
          if (i % 4 == 3) {
            rebase_if(!back);
          }
          st16(&M[(t + i + 1) * 8], m);
          if (i % 4 == 2 && (i > 2 || t > 0)) {
            rebase_if(back);
          }
        }
      }
      for (; t < T; t++) { // (3, 7 or 11 steps)
        const s2 x = ld(&Xp[t]), y = ld(&Yp[t]);
        if (t % 3 == 0) {
          step(std::integral_constant<int, 0>{}, x, y);
        } else if (t % 3 == 1) {
          step(std::integral_constant<int, 1>{}, x, y);
        } else {
          step(std::integral_constant<int, 2>{}, x, y);
        }
        if (t % 4 == 3) {
          rebase_if(!back);
        }
        st16(&M[(t + 1) * 8], m);
        if (t % 4 == 2 && t > 2) {
          rebase_if(back);
        }
      }
    }
    __syncthreads();

    // ---- the outputs (:150-196): step j from row j of the forward metrics and row K + 2 - j of the backward ones (beta(j + 1)); all 64 lanes, block after block
    for (int bb = 0; bb < n_live; bb++) {
      short*       Bb  = lds + (size_t)bb * kArrays * L;
      const short* ATb = Bb + (size_t)aAT * L;
      const short* BTb = Bb + (size_t)aBTR * L;
      const short* Xb  = Bb + (size_t)(dec1 ? (has_app ? aXA : aS) : aA2) * L;
      const short* Yb  = Bb + (size_t)(dec1 ? aP0 : aP1) * L;
      short*       Ob  = Bb + (size_t)(dec1 ? aE1 : aE2) * L;
      auto         pass = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        for (uint32_t j = 3u * (uint32_t)lane + R; j < K; j += 192) {
          const uint4 av = *reinterpret_cast<const uint4*>(ATb + (size_t)j * 8);
          const uint4 bv = *reinterpret_cast<const uint4*>(BTb + (size_t)(K + 2 - j) * 8);
          Ob[j]          = llr_of_step<R>(av, bv, ld(&Xb[j]), ld(&Yb[j]));
        }
      };
      pass(std::integral_constant<int, 0>{});
      pass(std::integral_constant<int, 1>{});
      pass(std::integral_constant<int, 2>{});
    }
    __syncthreads();
    if (!dec1) {
      for (uint32_t i = l16; i < K; i += 16) {
        A1[inter[i]] = E2[i];
      }
      __syncthreads();
    }
    n_run = n + 1;
    if (p.crc_poly) {
      // decode_tb_cb (sch.c:420-454): the checksum of the K hard bits after every half iteration (crc.c:92-140: MSB first, zero start; zero = good).
      // Eight lanes take an eighth of the bits each (both halves of the block's 16 do the same work) and shift their remainders into place
      // with x^(bits behind) mod g (p.crc_mult8, turbo_host.cpp).  A block that is good files its decision at once: its lanes walk on with the
      // others of the wave (the output pass is the whole wave's), on arrays nobody reads any more.
      const short*   Cs = (n_run & 1) ? E1 : A1;
      const uint32_t c8 = (K + 7) >> 3, lo = q * c8 < K ? q * c8 : K, hi = lo + c8 < K ? lo + c8 : K;
      uint32_t       c  = 0;
      for (uint32_t i = lo; i < hi; i++) {
        const uint32_t x = Cs[i] > 0 ? 1u : 0u;
        c = ((c << 1) & 0xffffffu) ^ ((((c >> 23) ^ x) & 1u) ? g24 : 0u);
      }
      const uint32_t mu = p.crc_mult8[q];
      uint32_t       r  = 0;
#pragma unroll 4
      for (int i = 23; i >= 0; i--) {
        r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? g24 : 0u);
        r ^= ((mu >> i) & 1u) ? c : 0u;
      }
      r ^= partner<0>(r);
      r ^= partner<1>(r);
      r ^= partner<2>(r);
      if (r == 0 && !done) {
        done     = true;
        crc_good = true;
        n_done   = n_run;
        if (live) {
          decide(n_run);
        }
      }
      if (__syncthreads_and(done || !live)) {
        break;
      }
    }
  }
  if (!done) {
    n_done = n_run;
    if (live) {
      decide(n_run);
    }
  }
  if (live) {
    if (l16 == 0) {
      if (p.noi) {
        p.noi[cb] = (int)(n_done - p.n_begin);
      }
      if (p.crc_ok) {
        p.crc_ok[cb] = crc_good ? 1 : 0;
      }
    }
    if (!p.desc) { // (a transport-block launch is never resumed)
      for (uint32_t i = l16; i < 7 * L; i += 16) {
        st[i] = B[i];
      }
    }
  }
}

} // namespace lat

size_t gen_lat_lds_bytes(uint32_t K)
{
  return (size_t)4 * lat::kArrays * (K + 4) * sizeof(short);
}

hipError_t launch_gen_lat(const GenParams& p, hipStream_t stream)
{
  const size_t lds = gen_lat_lds_bytes(p.K);
  if (lds > 156 * 1024 || (p.crc_poly && !p.crc_mult8) || (p.K & 7u)) {
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
  hipLaunchKernelGGL(lat::tdec_gen_lat_kernel, dim3(ceil_div(p.n_cb, 4)), dim3(64), lds, stream, p);
  return hipGetLastError();
}

} // namespace turbo
} // namespace phyhip
