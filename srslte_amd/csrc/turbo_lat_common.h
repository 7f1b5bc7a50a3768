// turbo_lat_common.h -- the in-place trellis butterflies of the latency kernels (turbo_lat_kernels.hip: the window decoders;
// turbo_gen_lat_kernels.hip: the scalar decoder): one trellis STATE per lane, 8 lanes per recursion, partners through DPP.  Device code only.
#pragma once
#include "turbo_arith.h"

#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace turbo {
namespace lat {

// ---- slot labelling: state held by slot s when r = step index mod 3
__host__ __device__ constexpr int state_of(int r, int s)
{
  const int x0 = s & 1, x1 = (s >> 1) & 1, x2 = (s >> 2) & 1;
  r %= 3;
  return r == 0 ? s : (r == 1 ? ((x0 << 2) | (x2 << 1) | x1) : ((x1 << 2) | (x0 << 1) | x2));
}
__host__ __device__ constexpr int slot_of(int r, int state) // inverse of state_of
{
  for (int s = 0; s < 8; s++) {
    if (state_of(r, s) == state) {
      return s;
    }
  }
  return 0;
}
// predecessor on the data-bit-0 branch of new state i (turbodecoder_win.h:753-790: m_b[i]); the data-bit-1 branch comes from PM[i] ^ 1
__host__ __device__ constexpr int pm_of(int i)
{
  constexpr int PM[8] = {0, 3, 4, 7, 1, 2, 5, 6};
  return PM[i];
}
// new states whose branches carry the parity LLR as {y, x} (the others: {0, x + y})
__host__ __device__ constexpr bool type_y(int i)
{
  return i == 1 || i == 2 || i == 5 || i == 6;
}

// invariants of the rotating labelling, for every residue r and slot s: the partner slot (s ^ 1, ^ 2, ^ 4) holds the state that differs in the
// shift register's outgoing bit; after the step both slots hold the two successors (u b2 b1) of that pair; three steps restore the labelling
__host__ __device__ constexpr bool labelling_ok()
{
  for (int r = 0; r < 3; r++) {
    for (int s = 0; s < 8; s++) {
      const int part = s ^ (1 << r), st = state_of(r, s), nx = state_of(r + 1, s);
      if (state_of(r, part) != (st ^ 1) || (nx & 3) != (st >> 1) || state_of(r + 1, part) != (nx ^ 4) || slot_of(r, st) != s) {
        return false;
      }
      if (pm_of(nx) != st && (pm_of(nx) ^ 1) != st) { // the slot's own state is one of the two predecessors of its new state
        return false;
      }
    }
  }
  return state_of(3, 5) == state_of(0, 5);
}
static_assert(labelling_ok(), "slot labelling of the in-place trellis butterflies");

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); // (no `old` operand to initialise)
}
// metric of the partner slot for a step at residue R: slot ^ 1, ^ 2, ^ 4
template <int R>
__device__ __forceinline__ uint32_t partner(uint32_t v)
{
  if constexpr (R == 0) {
    return dpp<0xB1>(v); // quad_perm [1,0,3,2]
  } else if constexpr (R == 1) {
    return dpp<0x4E>(v); // quad_perm [2,3,0,1]
  } else {
    // swap the two quads of every 8 lanes: banks 0 / 2 take from the lane 4 above, banks 1 / 3 from the lane 4 below
    int t = __builtin_amdgcn_mov_dpp((int)v, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    t     = __builtin_amdgcn_update_dpp(t, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xA, false);
    return (uint32_t)t;
  }
}
// value of slot 0 of every 8 lanes, in all 8
__device__ __forceinline__ uint32_t bcast_slot0(uint32_t v)
{
  const int t = __builtin_amdgcn_mov_dpp((int)v, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, true);
  return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x114 /* row_shr:4 */, 0xf, 0xA, false);
}
// maximum over the 8 slots, in all 8
__device__ __forceinline__ s2 max8(s2 v)
{
  v = vmax(v, from_u(partner<0>(to_u(v))));
  v = vmax(v, from_u(partner<1>(to_u(v))));
  v = vmax(v, from_u(partner<2>(to_u(v))));
  return v;
}

template <class AR>
__device__ __forceinline__ s2 normalise(s2 o)
{
  if constexpr (AR::kIs8) {
    return __builtin_elementwise_sub_sat(o, max8(o)); // turbodecoder_win.h:480-498, 8-bit: re-base on the maximum
  } else {
    return AR::sub(o, from_u(bcast_slot0(to_u(o)))); // 16-bit: subtract the metric of state 0 (slot 0 holds state 0 at every residue)
  }
}

// lane constants of a residue: this lane's new state carries {y, x}; this lane's own metric feeds the data-bit-0 branch
struct LaneK {
  bool ty[3], d0[3];
  bool q[3]; // LLR reduction: this lane keeps its own branch (odd slots collect the data-bit-1 maximum, even slots the data-bit-0 one)
  bool odd;
};
__device__ __forceinline__ LaneK lane_consts(int slot)
{
  LaneK c;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    bool ty = false, d0 = false;
#pragma unroll
    for (int s = 0; s < 8; s++) {
      const int  i  = state_of(r + 1, s);
      const bool t  = type_y(i);
      const bool d  = state_of(r, s) == pm_of(i);
      ty            = (slot == s) ? t : ty;
      d0            = (slot == s) ? d : d0;
    }
    c.ty[r] = ty;
    c.d0[r] = d0;
    c.q[r]  = (slot & 1) ? !d0 : d0;
  }
  c.odd = (slot & 1) != 0;
  return c;
}

// the two branch metrics of a lane for step operands (x, y) at residue R: own -> own new state, partner -> own new state (= own -> partner's)
template <class AR, int R>
__device__ __forceinline__ void gammas(const LaneK& c, s2 x, s2 y, s2& g_own, s2& g_cross)
{
  const s2 xy = AR::add(x, y);
  const s2 p  = c.ty[R] ? y : splat(0); // data bit 0
  const s2 q  = c.ty[R] ? x : xy;       // data bit 1
  g_own       = c.d0[R] ? p : q;
  g_cross     = c.d0[R] ? q : p;
}

// one backward step at residue R: beta(k + 1) in the labelling of R + 1 -> beta(k) in the labelling of R (turbodecoder_win.h:626-652)
template <class AR, int R>
__device__ __forceinline__ s2 beta_step(s2 b, s2 g_own, s2 g_cross)
{
  const s2 pb = from_u(partner<R>(to_u(b)));
  return AR::clean(vmax(AR::add_raw(b, g_own), AR::add_raw(pb, g_cross)));
}

// one forward step at residue R (turbodecoder_win.h:753-826); WITH_LLR: also max1 - max0 against beta(k + 1) `bn`
template <class AR, int R, bool WITH_LLR>
__device__ __forceinline__ s2 alpha_step(const LaneK& c, s2& a, s2 bn, s2 g_own, s2 g_cross)
{
  const s2 pa   = from_u(partner<R>(to_u(a)));
  const s2 t_o  = AR::add_raw(a, g_own);
  const s2 t_c  = AR::add_raw(pa, g_cross);
  s2       out  = splat(0);
  if constexpr (WITH_LLR) {
    // max over the 8 states of (branch + beta) for data bit 0 and for data bit 1.  Even slots collect the bit-0 maximum, odd slots the bit-1
    // one: a lane keeps the candidate of its class and sends the other to its slot ^ 1 partner, after which ONE value per lane is reduced
    // over slot ^ 2 and slot ^ 4 (11 instructions instead of the 18 of two full 8-lane reductions)
    const s2 mine  = c.q[R] ? t_o : t_c;
    const s2 other = c.q[R] ? t_c : t_o;
    s2       w     = vmax(AR::add_raw(bn, mine), from_u(partner<0>(to_u(AR::add_raw(bn, other)))));
    w              = vmax(w, from_u(partner<1>(to_u(w))));
    w              = vmax(w, from_u(partner<2>(to_u(w))));
    const s2 v     = from_u(partner<0>(to_u(w))); // the other class' maximum
    const s2 m1    = c.odd ? w : v;
    const s2 m0    = c.odd ? v : w;
    out            = AR::llr(AR::clean(m1), AR::clean(m0));
  }
  a = AR::clean(vmax(t_o, t_c));
  return out;
}

// the forward step alone, keeping the two branch sums (own -> own new state, partner -> own new state) for the LLR
template <class AR, int R>
__device__ __forceinline__ void alpha_branches(s2& a, s2 g_own, s2 g_cross, s2& t_o, s2& t_c)
{
  const s2 pa = from_u(partner<R>(to_u(a)));
  t_o         = AR::add_raw(a, g_own);
  t_c         = AR::add_raw(pa, g_cross);
  a           = AR::clean(vmax(t_o, t_c));
}

__device__ __forceinline__ void load8(const uint32_t* q, uint32_t (&r)[8])
{
  const uint4 a = *reinterpret_cast<const uint4*>(q), c = *reinterpret_cast<const uint4*>(q + 4);
  r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = c.x, r[5] = c.y, r[6] = c.z, r[7] = c.w;
}

} // namespace lat
} // namespace turbo
} // namespace phyhip
