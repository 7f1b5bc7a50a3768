// turbo_kernels.hip -- LTE turbo decoder (max-log-MAP SISO + QPP interleaver) for gfx950.
//
// Bit-exact re-design of the decoders srsran_tdec_run_all() dispatches to on an AVX2 host
// (reference: lib/src/phy/fec/turbo/turbodecoder.c:381-408):
//   * window decoder, 16 sub-blocks  (turbodecoder_win.h, WINIMP_IS_AVX16)  K%16==0 && K>800
//   * window decoder,  8 sub-blocks  (turbodecoder_win.h, WINIMP_IS_SSE16)  K%8==0  && K>400
//   * scalar decoder                 (turbodecoder_gen.c)                   otherwise
//
// MI355X mapping (not the SIMD layout of the reference):
//   * one lane owns TWO adjacent sub-blocks of one code block, packed as int16x2 in one VGPR; all 8
//     trellis states of both live in 8 VGPRs -> the ACS recursion is v_pk_add_i16(clamp)/v_pk_max_i16
//     with no cross-lane traffic; a code block is 8 lanes (16 sub-blocks) or 4 lanes (8 sub-blocks),
//     a wave decodes 8 or 16 code blocks in lock step.
//   * state metrics of the backward recursion are NOT stored for every step (that is 16 B per
//     trellis step, 196 KB of HBM traffic per half iteration at K=6144): the beta pass keeps one
//     check-point per 8 steps and the forward pass re-derives the 7 steps in between into registers
//     (bit-identical because the normalisation schedule is replayed exactly).
//   * per-code-block vectors live in HBM in a blocked layout [step/8][lane][step%8] of int16x2, so a
//     lane fetches 8 trellis steps with two dwordx4 loads and 8 lanes of a block read 256 contiguous
//     bytes.
//   * extrinsic exchange (turbodecoder_iter.h:104-128) is fused into the forward pass: a-priori
//     subtraction on load, QPP scatter on store.
#include "hip_common.h"
#include "turbo_arith.h"
#include "turbo_device.h"

namespace phyhip {
namespace turbo {

// Non-temporal workspace loads (NT): nothing the fixed-iteration 16-bit decoder reads is read again before several hundred KB per
// wave have passed, so keeping it in L2 only evicts lines that are still being written.  Measured on one box (K = 6144, 65,520 blocks,
// 8 half iterations): 12.03 ms without, 11.81 ms with the systematic / parity operands non-temporal, 11.74 ms with the exchanged rows
// too, 11.58 ms with the check-points as well; non-temporal STORES cost (12.2 ms).  The 8-bit decoders (14.7 -> 15.8 ms) and the
// early-stop mode (-2.5 % on the transport-block benches) lose with it -- their smaller, partly re-touched working sets do hit in L2 --
// so NT = fixed iterations and int16 only.
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ws_load16(const void* p)
{
  if constexpr (NT) {
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}

// one backward step, turbodecoder_win.h:626-652
template <class AR>
__device__ __forceinline__ void beta_step(s2 (&o)[8], s2 x, s2 y)
{
  auto adds = [](s2 a, s2 b) { return AR::add_raw(a, b); };
  s2   xy   = AR::add(x, y);
  s2 n0 = vmax(adds(o[4], xy), o[0]);
  s2 n1 = vmax(o[4], adds(o[0], xy));
  s2 n2 = vmax(adds(o[5], y), adds(o[1], x));
  s2 n3 = vmax(adds(o[5], x), adds(o[1], y));
  s2 n4 = vmax(adds(o[6], x), adds(o[2], y));
  s2 n5 = vmax(adds(o[6], y), adds(o[2], x));
  s2 n6 = vmax(o[7], adds(o[3], xy));
  s2 n7 = vmax(adds(o[7], xy), o[3]);
  o[0] = AR::clean(n0);
  o[1] = AR::clean(n1);
  o[2] = AR::clean(n2);
  o[3] = AR::clean(n3);
  o[4] = AR::clean(n4);
  o[5] = AR::clean(n5);
  o[6] = AR::clean(n6);
  o[7] = AR::clean(n7);
}

// one forward step, turbodecoder_win.h:753-826.  WITH_LLR: also max1-max0 using the beta of the next step.
template <class AR, bool WITH_LLR>
__device__ __forceinline__ s2 alpha_step(s2 (&o)[8], const s2 (&b)[8], s2 x, s2 y)
{
  auto adds = [](s2 a, s2 b) { return AR::add_raw(a, b); };
  s2   xy   = AR::add(x, y);
  s2 m_b[8], nw[8];
  m_b[0] = o[0];
  m_b[1] = adds(o[3], y);
  m_b[2] = adds(o[4], y);
  m_b[3] = o[7];
  m_b[4] = o[1];
  m_b[5] = adds(o[2], y);
  m_b[6] = adds(o[5], y);
  m_b[7] = o[6];
  nw[0] = adds(o[1], xy);
  nw[1] = adds(o[2], x);
  nw[2] = adds(o[5], x);
  nw[3] = adds(o[6], xy);
  nw[4] = adds(o[0], xy);
  nw[5] = adds(o[3], x);
  nw[6] = adds(o[4], x);
  nw[7] = adds(o[7], xy);
  s2 out = splat(0);
  if (WITH_LLR) {
    s2 m0 = adds(b[0], m_b[0]);
    s2 m1 = adds(b[0], nw[0]);
#pragma unroll
    for (int i = 1; i < 8; i++) {
      m0 = vmax(m0, adds(b[i], m_b[i]));
      m1 = vmax(m1, adds(b[i], nw[i]));
    }
    out = AR::llr(AR::clean(m1), AR::clean(m0));
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    o[i] = AR::clean(vmax(m_b[i], nw[i]));
  }
  return out;
}

__device__ __forceinline__ short wrap16(int v)
{
  return (short)v;
}

// Blocked arrays hold, per 8-step block and lane, 8 dwords.  They are stored as two half-blocks of 4 dwords so
// that each dwordx4 access of a wave covers one contiguous 1 KB (measured: 5.5 TB/s against 4.5 TB/s for a
// 32-byte-per-lane layout where every 128-byte line is touched by two instructions).
template <bool NT = false>
__device__ __forceinline__ void load_block(const uint32_t* arr, uint32_t blk_lane, uint32_t (&r)[8])
{
  const uint32_t blk = blk_lane >> 6, ln = blk_lane & 63u;
  const uint4    a = ws_load16<NT>(arr + ((size_t)(blk * 2) * 64 + ln) * 4);
  const uint4    c = ws_load16<NT>(arr + ((size_t)(blk * 2 + 1) * 64 + ln) * 4);
  r[0] = a.x;
  r[1] = a.y;
  r[2] = a.z;
  r[3] = a.w;
  r[4] = c.x;
  r[5] = c.y;
  r[6] = c.z;
  r[7] = c.w;
}

__device__ __forceinline__ void store_block(uint32_t* arr, uint32_t blk_lane, const uint32_t (&r)[8])
{
  const uint32_t blk = blk_lane >> 6, ln = blk_lane & 63u;
  *reinterpret_cast<uint4*>(arr + ((size_t)(blk * 2) * 64 + ln) * 4)     = make_uint4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<uint4*>(arr + ((size_t)(blk * 2 + 1) * 64 + ln) * 4) = make_uint4(r[4], r[5], r[6], r[7]);
}

// exchange tables: 8 dwords per (block, lane of the code block), contiguous
__device__ __forceinline__ void load_lut(const uint32_t* arr, uint32_t idx, uint32_t (&r)[8])
{
  const uint4* q = reinterpret_cast<const uint4*>(arr + (size_t)idx * 8);
  const uint4  a = q[0], c = q[1];
  r[0] = a.x;
  r[1] = a.y;
  r[2] = a.z;
  r[3] = a.w;
  r[4] = c.x;
  r[5] = c.y;
  r[6] = c.z;
  r[7] = c.w;
}

// Blocked int16 index of trellis step k of sub-block d (LPC lanes per code block)
template <int LPC>
__host__ __device__ __forceinline__ uint32_t elem_index(uint32_t k, uint32_t d)
{
  return ((((k >> 3) * LPC + (d >> 1)) * 8 + (k & 7)) << 1) + (d & 1);
}

// ------------------------------------------------------------------------------------------------
// Windowed decoder kernel: one wave per 64/LPC code blocks, whole srsran_tdec_run_all in one launch.
//
// Per-code-block workspace (dwords, AW = nblk*8*LPC each):
//   S, P0, P1 : systematic / parity LLRs, blocked layout [step/8][lane][step%8]   (read only)
//   A1        : a-priori of decoder 1, ALREADY  app1 - ext1  (turbodecoder_iter.h:108), row layout [step][lane]
//   E1        : ext1 of decoder 1 (after the subtraction of :115 once n >= 2), row layout
//   A2        : input of decoder 2 (interleaved ext1), row layout
//   CK        : backward-recursion check-points, one per 8 steps
// The QPP interleaver is contention free for both directions: the 16 (8) sub-block outputs of one
// trellis step land in ONE row of the destination, permuted.  So the extrinsic exchange is, per step,
// one in-register permutation across the lanes of the code block (ds_bpermute) and one full 32-byte
// row store -- never a 2-byte scatter.
// ------------------------------------------------------------------------------------------------
// rows of the 64/LPC code blocks of a wave are interleaved: row k of the wave is 64 contiguous dwords
__device__ __forceinline__ void load_rows(const uint32_t* arr, uint32_t b, int lane, uint32_t (&r)[8])
{
  const uint32_t* q = arr + (size_t)(b * 8) * 64 + lane;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    r[j] = q[j * 64];
  }
}

// The same 8 rows fetched with TWO dwordx4 per lane (the 2 KB of rows 8b..8b+7 are contiguous): dword-per-lane
// loads top out near 3 TB/s on this part, 16-byte ones reach 5.5 TB/s.  Lane L then holds columns 4(L%16)..+3
// of rows L/16 and 4 + L/16; rows_to_lane() turns that into "8 rows of column L" through a 2 KB LDS image.
template <bool NT = false>
__device__ __forceinline__ void issue_rows(const uint32_t* arr, uint32_t b, int lane, uint32_t (&t)[8])
{
  const uint4* q = reinterpret_cast<const uint4*>(arr + (size_t)(b * 8) * 64) + lane;
  const uint4  a = ws_load16<NT>(q), c = ws_load16<NT>(q + 64);
  t[0] = a.x;
  t[1] = a.y;
  t[2] = a.z;
  t[3] = a.w;
  t[4] = c.x;
  t[5] = c.y;
  t[6] = c.z;
  t[7] = c.w;
}

__device__ __forceinline__ void rows_to_lane(uint32_t* stage, int lane, const uint32_t (&t)[8], uint32_t (&r)[8])
{
  // one wave per workgroup and the LDS pipeline is in order: no barrier between the write and the read
  reinterpret_cast<uint4*>(stage)[lane]      = make_uint4(t[0], t[1], t[2], t[3]);
  reinterpret_cast<uint4*>(stage)[64 + lane] = make_uint4(t[4], t[5], t[6], t[7]);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    r[j] = stage[j * 64 + lane];
  }
}

// ---- storage policy.  The 16-bit decoders keep one int16x2 dword per (lane, step) in the workspace.  The 8-bit decoders' values are
// int8 by construction (in registers: value << 8 in each int16 half, low bytes zero -- see Ar8), so their workspace holds ONE 16-bit
// word per (lane, step): half the HBM traffic of a kernel that runs at the HBM ceiling.  S8 = AR::kIs8 selects the layout:
//   blocked arrays (S, P0, P1, check-points): 8 steps of a lane = 16 bytes = ONE dwordx4 (1 KB contiguous per wave instruction)
//   row arrays (A1, A2, D): a row of the wave = 64 x 2 bytes; the 8 rows of a block are 1 KB = ONE dwordx4 per lane
// Loads stay PACKED in the prefetch registers (4 dwords instead of 8 per operand block) and are widened where they are consumed:
// one v_perm_b32 per step puts the two bytes into the high bytes of the halves; one v_perm_b32 packs two steps for a store.
__device__ __forceinline__ uint32_t s8_unpack_lo(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x010c000cu); } // bytes 0, 1
__device__ __forceinline__ uint32_t s8_unpack_hi(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x030c020cu); } // bytes 2, 3
__device__ __forceinline__ uint32_t s8_pack2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07050301u); } // (a.b1, a.b3, b.b1, b.b3)
__device__ __forceinline__ uint16_t s8_pack1(uint32_t a) { return (uint16_t)__builtin_amdgcn_perm(0u, a, 0x0c0c0301u); }

// raw (as stored) form of one 8-step block of a blocked array: 8 dwords, or 4 with 8-bit storage
template <bool S8, bool NT = false>
__device__ __forceinline__ void load_block_raw(const uint32_t* arr, uint32_t blk_lane, uint32_t (&r)[8])
{
  if constexpr (S8) {
    const uint4 a = ws_load16<NT>(arr + (size_t)blk_lane * 4);
    r[0] = a.x;
    r[1] = a.y;
    r[2] = a.z;
    r[3] = a.w;
  } else {
    load_block<NT>(arr, blk_lane, r);
  }
}
// raw -> the 8 int16x2 values of the block
template <bool S8>
__device__ __forceinline__ void block_values(const uint32_t (&raw)[8], uint32_t (&v)[8])
{
#pragma unroll
  for (int d = 0; d < 4; d++) {
    v[2 * d]     = S8 ? s8_unpack_lo(raw[d]) : raw[2 * d];
    v[2 * d + 1] = S8 ? s8_unpack_hi(raw[d]) : raw[2 * d + 1];
  }
}
template <bool S8>
__device__ __forceinline__ void store_block_v(uint32_t* arr, uint32_t blk_lane, const uint32_t (&v)[8])
{
  if constexpr (S8) {
    *reinterpret_cast<uint4*>(arr + (size_t)blk_lane * 4) = make_uint4(s8_pack2(v[0], v[1]), s8_pack2(v[2], v[3]), s8_pack2(v[4], v[5]), s8_pack2(v[6], v[7]));
  } else {
    store_block(arr, blk_lane, v);
  }
}
// the 8 rows of block b of a row array, as stored: two dwordx4 per lane, or one with 8-bit storage
template <bool S8, bool NT = false>
__device__ __forceinline__ void issue_rows_raw(const uint32_t* arr, uint32_t b, int lane, uint32_t (&t)[8])
{
  if constexpr (S8) {
    const uint4 a = ws_load16<NT>(reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(arr) + (size_t)(b * 8) * 64) + lane);
    t[0] = a.x;
    t[1] = a.y;
    t[2] = a.z;
    t[3] = a.w;
  } else {
    issue_rows<NT>(arr, b, lane, t);
  }
}
// ... turned into "8 rows of this lane's column" (int16x2 values) through the LDS stage
template <bool S8>
__device__ __forceinline__ void rows_to_lane_v(uint32_t* stage, int lane, const uint32_t (&t)[8], uint32_t (&r)[8])
{
  if constexpr (S8) {
    reinterpret_cast<uint4*>(stage)[lane] = make_uint4(t[0], t[1], t[2], t[3]); // in-order LDS pipeline, one wave per workgroup: no barrier
    const uint16_t* s16 = reinterpret_cast<const uint16_t*>(stage);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      r[j] = s8_unpack_lo((uint32_t)s16[j * 64 + lane]);
    }
  } else {
    rows_to_lane(stage, lane, t, r);
  }
}
// one element of a row array: (row, lane) <- int16x2 value
template <bool S8>
__device__ __forceinline__ void store_row(uint32_t* arr, size_t row, int lane, uint32_t v)
{
  if constexpr (S8) {
    reinterpret_cast<uint16_t*>(arr)[row * 64 + lane] = s8_pack1(v);
  } else {
    arr[row * 64 + lane] = v;
  }
}

// value for this lane's two destination sub-blocks, fetched from the lanes holding the source sub-blocks
template <int LPC>
__device__ __forceinline__ uint32_t permute_pair(uint32_t v, uint32_t sel)
{
  const uint32_t jlo = sel & 31u, jhi = (sel >> 5) & 31u;
  const uint32_t a   = __shfl(v, (int)(jlo >> 1), LPC);
  const uint32_t c   = __shfl(v, (int)(jhi >> 1), LPC);
  const uint32_t lo  = (jlo & 1u) ? (a >> 16) : (a & 0xffffu);
  const uint32_t hi  = (jhi & 1u) ? (c >> 16) : (c & 0xffffu);
  return lo | (hi << 16);
}

// phase 0: input extraction (turbodecoder_win.h:888-930 / turbodecoder_iter.h:58-70,88-102) from int16 or int8
// LLRs.  All 48 element loads of an 8-step block are issued before the first use (addresses clamped instead
// of branching on the ragged last block), so the block costs one memory round trip, not eight.
template <int LPC, class AR, typename T>
__device__ __forceinline__ void extract_input(const T* in, int sb_layout, uint32_t K, uint32_t long_sb, uint32_t nblk,
                                              int lane, int pl, uint32_t* S, uint32_t* P0, uint32_t* P1, short* TL,
                                              uint32_t b_first = 0, bool tails = true)
{
  constexpr int NB = 2 * LPC;
  for (uint32_t b = b_first; b < nblk; b++) {
    const int nv = (int)(long_sb - b * 8) < 8 ? (int)(long_sb - b * 8) : 8; // valid steps in this block
    short     r[2][24];
    if (sb_layout) {
      // rm_turbo layout: element (step k, sub-block d) of array a at in[a*(K+32) + k*NB + d]
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t k = b * 8 + (j < nv ? j : nv - 1);
#pragma unroll
        for (int a3 = 0; a3 < 3; a3++) {
          r[0][3 * j + a3] = AR::conv_in(in[a3 * (K + 32) + k * NB + 2 * pl]);
          r[1][3 * j + a3] = AR::conv_in(in[a3 * (K + 32) + k * NB + 2 * pl + 1]);
        }
      }
    } else {
      // natural order: the 8 steps of one sub-block are 24 consecutive LLRs [s p0 p1]...
      const T*  c0  = in + 3 * ((size_t)(2 * pl) * long_sb + b * 8);
      const T*  c1  = in + 3 * ((size_t)(2 * pl + 1) * long_sb + b * 8);
      const int lim = 3 * nv - 1;
#pragma unroll
      for (int t = 0; t < 24; t++) {
        const int tt = t < lim ? t : lim;
        r[0][t]      = AR::conv_in(c0[tt]);
        r[1][t]      = AR::conv_in(c1[tt]);
      }
    }
    uint32_t s[8], y0[8], y1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      s[j]  = (uint32_t)(uint16_t)r[0][3 * j] | ((uint32_t)(uint16_t)r[1][3 * j] << 16);
      y0[j] = (uint32_t)(uint16_t)r[0][3 * j + 1] | ((uint32_t)(uint16_t)r[1][3 * j + 1] << 16);
      y1[j] = (uint32_t)(uint16_t)r[0][3 * j + 2] | ((uint32_t)(uint16_t)r[1][3 * j + 2] << 16);
    }
    store_block_v<AR::kIs8>(S, b * 64 + lane, s);
    store_block_v<AR::kIs8>(P0, b * 64 + lane, y0);
    store_block_v<AR::kIs8>(P1, b * 64 + lane, y1);
  }
  if (pl == 0 && tails) {
    const uint32_t tb = sb_layout ? 3 * (K + 32) : 3 * K;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      TL[i]     = AR::conv_in(in[tb + 2 * i]);         // syst tail
      TL[3 + i] = AR::conv_in(in[tb + 2 * i + 1]);     // parity0 tail
      TL[6 + i] = AR::conv_in(in[tb + 6 + 2 * i]);     // app2 tail
      TL[9 + i] = AR::conv_in(in[tb + 6 + 2 * i + 1]); // parity1 tail
    }
  }
}

// Fast input extraction for natural-order int16 LLRs [s p0 p1]xK (what srsran_tdec_run_all gets with
// srsran_tdec_force_not_sb): the stream is sub-block major (the 3W LLRs of a sub-block are contiguous) while the
// decoder wants step-major data spread over lanes, so every code block of the wave is staged through LDS in chunks of
// NBK 8-step blocks: the 64 lanes copy the NB contiguous runs of 48*NBK bytes with 8-byte loads (each wave-level load
// covers >= 256 contiguous bytes), then lane (p', g) assembles the blocked dwords of sub-block pair p' for two blocks
// and stores them into the slots of the lane that owns that pair.  Needs W % 4 == 0 and 8-byte aligned code blocks.
template <int LPC, class AR>
__device__ __forceinline__ void extract_input_natural16(const short* in_wave, uint32_t in_stride, int n_cb_left, uint32_t K,
                                                        uint32_t long_sb, uint32_t nblk, int lane, uint32_t* S, uint32_t* P0,
                                                        uint32_t* P1, short* TL_wave, uint2* stage)
{
  constexpr int NB  = 2 * LPC;
  constexpr int CPW = 64 / LPC;
  constexpr int NBK = 256 / NB;  // blocks per chunk: NB runs of 48*NBK bytes = 12 KB of LDS
  constexpr int RS  = 6 * NBK + 1; // run stride in the LDS image, in 8-byte units (+1: spreads the LDS banks)
  const int     pp  = lane % LPC, g = lane / LPC;
  // chunks are (code block, block range) pairs; the loads of the next chunk are in flight while the current one is
  // re-distributed (24 8-byte loads per lane and chunk: NB * 6 * NBK / 64)
  constexpr int NLD = NB * 6 * NBK / 64;
  const uint32_t nchunk = (nblk + NBK - 1) / NBK, total = CPW * nchunk;
  auto chunk_src = [&](uint32_t c, const short*& in, uint32_t& b0, int& nbt) {
    const int cw = (int)(c % CPW); // block range outermost: the 8 code blocks' 128-byte pieces of a 1 KB line are written back to back
    in           = in_wave + (size_t)(cw < n_cb_left ? cw : n_cb_left - 1) * in_stride;
    b0           = (c / CPW) * NBK;
    nbt          = (int)(nblk - b0) < NBK ? (int)(nblk - b0) : NBK;
  };
  auto issue_chunk = [&](uint32_t c, uint2(&rg)[NLD]) {
    const short* in;
    uint32_t     b0;
    int          nbt;
    chunk_src(c, in, b0, nbt);
    if (nbt == NBK) { // full chunk: the run length is a compile-time constant (no integer division by a variable)
      constexpr int rl = 6 * NBK;
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int i = t * 64 + lane, d = i / rl, o = i - d * rl;
        rg[t]       = *(reinterpret_cast<const uint2*>(in + 3 * ((size_t)d * long_sb + (size_t)b0 * 8)) + o);
      }
    } else {
      const int rl = 6 * nbt; // run length in 8-byte units
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int i = t * 64 + lane;
        if (i < NB * rl) {
          const int d = i / rl, o = i - d * rl;
          rg[t]       = *(reinterpret_cast<const uint2*>(in + 3 * ((size_t)d * long_sb + (size_t)b0 * 8)) + o);
        }
      }
    }
  };
  uint2 rg[NLD];
  issue_chunk(0, rg);
  for (uint32_t c = 0; c < total; c++) {
    const short* in;
    uint32_t     b0;
    int          nbt;
    chunk_src(c, in, b0, nbt);
    const int cw = (int)(c % CPW);
    if (nbt == NBK) {
      constexpr int rl = 6 * NBK;
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int i = t * 64 + lane, d = i / rl, o = i - d * rl;
        stage[d * RS + o] = rg[t];
      }
    } else {
      const int rl = 6 * nbt;
#pragma unroll
      for (int t = 0; t < NLD; t++) {
        const int i = t * 64 + lane;
        if (i < NB * rl) {
          const int d = i / rl, o = i - d * rl;
          stage[d * RS + o] = rg[t];
        }
      }
    }
    if (c + 1 < total) {
      issue_chunk(c + 1, rg);
    }
    {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int lb = g * 2 + h; // NBK * LPC / 64 == 2 blocks per lane
        if (lb < nbt) {
          short r[2][24];
#pragma unroll
          for (int dd = 0; dd < 2; dd++) {
            const uint2* q = stage + (2 * pp + dd) * RS + lb * 6;
#pragma unroll
            for (int t = 0; t < 6; t++) {
              const uint2 v    = q[t];
              r[dd][4 * t]     = (short)(v.x & 0xffffu);
              r[dd][4 * t + 1] = (short)(v.x >> 16);
              r[dd][4 * t + 2] = (short)(v.y & 0xffffu);
              r[dd][4 * t + 3] = (short)(v.y >> 16);
            }
          }
          uint32_t sv[8], y0[8], y1[8];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            sv[j] = (uint32_t)(uint16_t)AR::conv_in(r[0][3 * j]) | ((uint32_t)(uint16_t)AR::conv_in(r[1][3 * j]) << 16);
            y0[j] = (uint32_t)(uint16_t)AR::conv_in(r[0][3 * j + 1]) | ((uint32_t)(uint16_t)AR::conv_in(r[1][3 * j + 1]) << 16);
            y1[j] = (uint32_t)(uint16_t)AR::conv_in(r[0][3 * j + 2]) | ((uint32_t)(uint16_t)AR::conv_in(r[1][3 * j + 2]) << 16);
          }
          const uint32_t slot = (b0 + lb) * 64 + cw * LPC + pp;
          store_block_v<AR::kIs8>(S, slot, sv);
          store_block_v<AR::kIs8>(P0, slot, y0);
          store_block_v<AR::kIs8>(P1, slot, y1);
        }
      }
    }
  }
  // tail LLRs: lane cw * LPC of every code block
  if (pp == 0) {
    const short* in = in_wave + (size_t)(g < n_cb_left ? g : n_cb_left - 1) * in_stride;
    short*       TL = TL_wave + 16 * g;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      TL[i]     = AR::conv_in(in[3 * K + 2 * i]);
      TL[3 + i] = AR::conv_in(in[3 * K + 2 * i + 1]);
      TL[6 + i] = AR::conv_in(in[3 * K + 6 + 2 * i]);
      TL[9 + i] = AR::conv_in(in[3 * K + 6 + 2 * i + 1]);
    }
  }
}

// Fast input extraction for the rm_turbo sub-block layout (int16): element (step k, sub-block d) of stream a sits at
// in[a (K+32) + k NB + d], i.e. the 8 steps of a block are 8 * NB contiguous int16 per code block and stream.  The LPC
// lanes of a code block fetch them with two dwordx4 each (128 contiguous bytes per code block and instruction) and the
// [step][sub-block pair] image is turned into "8 steps of pair p" through the 2 KB LDS stage.  Handles the `nblk` full
// 8-step blocks it is given (a ragged last block goes through extract_input); needs 16-byte aligned code blocks.
template <int LPC, class AR>
__device__ __forceinline__ void extract_input_sb16(const short* in, uint32_t K, uint32_t nblk, int lane, int pl, uint32_t* S,
                                                   uint32_t* P0, uint32_t* P1, short* TL, uint32_t* stage)
{
  constexpr int  NB  = 2 * LPC;
  const int      cbw = lane / LPC;
  uint4*         st4 = reinterpret_cast<uint4*>(stage);
  // one stream (systematic / parity 0 / parity 1) of one 8-step block: two 16-byte pieces per lane through the staging image
  auto put = [&](uint32_t* dst, uint32_t b, const uint4& lo, const uint4& hi) {
    st4[cbw * 2 * LPC + pl]       = lo;
    st4[cbw * 2 * LPC + LPC + pl] = hi;
    uint32_t r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t w = stage[cbw * 8 * LPC + j * LPC + pl];
      r[j] = (uint32_t)(uint16_t)AR::conv_in((short)(w & 0xffffu)) | ((uint32_t)(uint16_t)AR::conv_in((short)(w >> 16)) << 16);
    }
    store_block_v<AR::kIs8>(dst, b * 64 + lane, r);
  };
  for (uint32_t b = 0; b < nblk; b++) {
    // all six loads of the block are issued before the first use (named registers: an indexed local array ends up in scratch)
    const uint4* q0 = reinterpret_cast<const uint4*>(in + (size_t)b * 8 * NB);
    const uint4* q1 = reinterpret_cast<const uint4*>(in + (size_t)(K + 32) + (size_t)b * 8 * NB);
    const uint4* q2 = reinterpret_cast<const uint4*>(in + (size_t)2 * (K + 32) + (size_t)b * 8 * NB);
    const uint4  s_lo = q0[pl], s_hi = q0[LPC + pl];
    const uint4  y_lo = q1[pl], y_hi = q1[LPC + pl];
    const uint4  z_lo = q2[pl], z_hi = q2[LPC + pl];
    put(S, b, s_lo, s_hi);
    put(P0, b, y_lo, y_hi);
    put(P1, b, z_lo, z_hi);
  }
  if (pl == 0) {
    const uint32_t tb = 3 * (K + 32);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      TL[i]     = AR::conv_in(in[tb + 2 * i]);
      TL[3 + i] = AR::conv_in(in[tb + 2 * i + 1]);
      TL[6 + i] = AR::conv_in(in[tb + 6 + 2 * i]);
      TL[9 + i] = AR::conv_in(in[tb + 6 + 2 * i + 1]);
    }
  }
}

// One unit of work: the CPW = 64 / LPC code blocks wb * CPW ... of the batch, decoded by one wave in the workspace slab `slab`.
// Bl: re-derived backward metrics of the current 8-step block (8 steps x 8 states x int16x2 per lane); Tr: staging image of 8
// exchanged rows (rows_to_lane).
// ES: early-stop / descriptor mode (transport-block decoding, sch_host.cpp).  A separate instantiation: the CRC state and
// the per-block descriptors must not cost the fixed-iteration kernel registers (it runs at 2 waves per SIMD).
template <int LPC, class AR, bool ES>
__device__ __forceinline__ void tdec_win_unit(const WinParams& p, const uint32_t wb, const uint32_t slab, const int lane, uint4 (&Bl)[8][2][64],
                                              uint32_t (&Tr)[512])
{
  const uint32_t crc_poly = ES ? p.crc_poly : 0u;
  const CbDesc*  desc     = ES ? p.desc : nullptr;
  constexpr int  NB  = 2 * LPC;
  constexpr int  CPW = 64 / LPC;
  constexpr bool NT  = !ES && !AR::kIs8; // non-temporal workspace loads (see ws_load16)
  const int     pl   = lane % LPC;
  const int     cb_raw = (int)wb * CPW + lane / LPC;
  // Lane groups past the end of the batch stay in the wave: the exchanged rows are loaded 16 bytes per lane and
  // re-distributed through LDS (issue_rows / rows_to_lane), which needs all 64 lanes.  They decode a copy of the last
  // code block into their own (allocated) workspace slots and write no output.
  const bool    live = cb_raw < p.n_cb;
  const int     cb   = live ? cb_raw : p.n_cb - 1;
  const uint32_t K       = p.K;
  const uint32_t long_sb = K / NB;
  const uint32_t nblk    = (long_sb + 7) >> 3;
  const uint32_t AW      = nblk * LPC * 8;

  // The workspace of the 64/LPC code blocks of this wave is ONE slab with the blocks interleaved at lane
  // granularity: every wave-level load/store touches one contiguous 256 B (dword) or 1 KB (dwordx4) run,
  // and the per-step row exchange of all the wave's code blocks is a single contiguous 256 B store.
  const uint32_t AWG = AW * CPW; // dwords per array per wave
  uint32_t* ws  = p.ws + (size_t)slab * p.ws_stride * CPW;
  uint32_t* S   = ws;
  uint32_t* P0  = ws + AWG;
  uint32_t* P1  = ws + 2 * AWG;
  uint32_t* A1  = ws + 3 * AWG;
  uint32_t* D   = ws + 4 * AWG; // decision LLRs of the last half iteration of a launch
  uint32_t* A2  = ws + 5 * AWG;
  uint32_t* CK  = ws + 6 * AWG;
  short*    TL  = reinterpret_cast<short*>(CK + (size_t)(nblk + 1) * 64 * 8) + 16 * (lane / LPC); // 12 tail LLRs per block

  // ---- phase 0: input extraction
  if (p.n_begin == 0) {
    // sub-blocks that are a multiple of 4 steps keep every run 8-byte aligned; a ragged last block (K = 5824: 364 = 45 * 8 + 4 steps) is
    // fetched whole -- the 4 steps behind it are the head of the next sub-block or the tail LLRs, inside the code block's
    // 3 K + 12 values, and the decoder never looks at steps >= long_sb
    const bool fast = !desc && !p.in_is8 && !p.sb_layout && (long_sb & 3u) == 0 && ((reinterpret_cast<uintptr_t>(p.input) | (2u * p.in_stride)) & 7u) == 0;
    if (fast) {
      const int first = (int)wb * CPW;
      extract_input_natural16<LPC, AR>(p.input + (size_t)first * p.in_stride, p.in_stride, p.n_cb - first, K, long_sb, nblk, lane,
                                       S, P0, P1, TL - 16 * (lane / LPC), reinterpret_cast<uint2*>(&Bl[0][0][0]));
    } else if (!p.in_is8 && p.sb_layout &&
               __all(((reinterpret_cast<uintptr_t>(p.input) + 2 * (desc ? (size_t)desc[cb].in_off : (size_t)cb * p.in_stride)) & 15u) == 0)) {
      // K = 5824 (the code block size of the largest 20 MHz transport blocks) has sub-blocks of 364 steps: 45 full blocks + 4 steps
      const short* in = p.input + (desc ? (size_t)desc[cb].in_off : (size_t)cb * p.in_stride);
      extract_input_sb16<LPC, AR>(in, K, long_sb >> 3, lane, pl, S, P0, P1, TL, Tr);
      if (long_sb & 7u) {
        extract_input<LPC, AR>(in, 1, K, long_sb, nblk, lane, pl, S, P0, P1, TL, long_sb >> 3, false);
      }
    } else if (p.in_is8) {
      const signed char* in = reinterpret_cast<const signed char*>(p.input) + (desc ? (size_t)desc[cb].in_off : (size_t)cb * p.in_stride);
      extract_input<LPC, AR>(in, p.sb_layout, K, long_sb, nblk, lane, pl, S, P0, P1, TL);
    } else {
      const short* in = p.input + (desc ? (size_t)desc[cb].in_off : (size_t)cb * p.in_stride);
      extract_input<LPC, AR>(in, p.sb_layout, K, long_sb, nblk, lane, pl, S, P0, P1, TL);
    }
  }
  // rows of the exchanged vectors whose subtraction wraps instead of saturating (8-bit only: the ragged
  // tail of srsran_vec_sub_bbb's 32-byte vector loop, vector_simd.c:162-190)
  const uint32_t wrap_row = (AR::kIs8 && (K & 31u)) ? long_sb - 1 : 0xffffffffu;
  __syncthreads();

  // ---- half iterations (turbodecoder_iter.h:72-141)
  // ---- hard decision (turbodecoder.c:370-378 + turbodecoder_win.h:973-993): bit = LLR > 0, MSB first.
  // Source: app1 after an even number of half iterations, else ext1; the last half iteration filed it in D.
  // With a CRC generator the checksum of the K hard bits is formed on the way (sch.c:430-447: zero means the code
  // block is good): every lane runs the bit-serial CRC of its two sub-blocks, the 16 partial checksums are shifted
  // to their place by multiplication with x^(W (NB-1-d)) mod g and XOR-ed across the lanes of the code block.
  uint8_t*       out       = p.output + (desc ? (size_t)desc[cb].out_off : (size_t)cb * p.out_stride);
  const uint32_t out_bytes = desc ? desc[cb].out_bytes : K / 8;
  // Early stop: a code block that has matched its CRC (or a lane group behind the end of the batch) stays in the wave until
  // the wave's last block is done, but its vectors are dead.  Its block / row indices are masked to 0 so that it keeps
  // re-touching one cached line per array instead of streaming its workspace through HBM: m_own covers the accesses a lane
  // makes for its own sub-blocks, m_row the 16-byte row pieces it fetches for the lane group that owns those columns.
  uint32_t m_own_v = ~0u, m_row_v = ~0u;
  auto     set_masks = [&](bool dead) {
    const unsigned long long dm = __ballot(dead);
    // first lane of the group owning the columns this lane fetches in issue_rows: 4 (lane % 16) .. + 3, or 8 (lane % 8) .. + 7 with 8-bit storage
    const int                og = AR::kIs8 ? ((8 * (lane & 7)) / LPC) * LPC : ((4 * (lane & 15)) / LPC) * LPC;
    m_own_v                     = dead ? 0u : ~0u;
    m_row_v                     = ((dm >> og) & 1ull) ? 0u : ~0u;
  };
  if (ES) {
    set_masks(!live);
  }
#define M_OWN (ES ? m_own_v : ~0u)
#define M_ROW (ES ? m_row_v : ~0u)
  auto decide = [&](bool write, bool final_try) -> uint32_t {
    short*         o16   = (p.dec_llr && live && write) ? p.dec_llr + (size_t)cb * K : nullptr;
    const bool     whole = (long_sb & 7) == 0;
    const uint32_t bps   = long_sb >> 3; // bytes per sub-block
    const uint32_t poly  = crc_poly & 0xffffffu;
    uint32_t       c0 = 0, c1 = 0;
    // ragged sub-blocks (long_sb not a multiple of 8): the hard bits of every sub-block are first packed MSB first into a
    // byte image in LDS (the beta buffer is idle between half iterations), sub-block d at sbuf[d][.] with a zero byte
    // behind it, and the natural-order bytes are cut out of that image afterwards
    const uint32_t sbs  = nblk + 1;
    uint8_t*       sbuf = reinterpret_cast<uint8_t*>(&Bl[0][0][0]) + (size_t)(lane / LPC) * NB * sbs;
    static_assert(sizeof(Bl) >= 64 / LPC * 2 * LPC * (6144 / (2 * LPC) / 8 + 2), "byte image does not fit");
    if (!whole) {
      sbuf[(2 * pl) * sbs + nblk]     = 0;
      sbuf[(2 * pl + 1) * sbs + nblk] = 0;
    }
    {
      const bool wide = whole && ((bps & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0) && out_bytes == K / 8;
      uint32_t   w0 = 0, w1 = 0;
      uint32_t   t[8], tn[8];
      issue_rows_raw<AR::kIs8, NT>(D, 0, lane, t);
      for (uint32_t b = 0; b < nblk; b++) {
        if (b + 1 < nblk) {
          issue_rows_raw<AR::kIs8, NT>(D, (b + 1) & M_ROW, lane, tn);
        }
        uint32_t r[8];
        rows_to_lane_v<AR::kIs8>(Tr, lane, t, r);
        uint32_t b0 = 0, b1 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (b * 8 + j < long_sb) {
            const s2       v  = from_u(r[j]);
            const uint32_t x0 = v.x > 0 ? 1u : 0u, x1 = v.y > 0 ? 1u : 0u;
            b0 |= (x0 << 7) >> j;
            b1 |= (x1 << 7) >> j;
            if (crc_poly) { // crc.c:92-140, MSB first, zero initial state
              c0 = ((c0 << 1) & 0xffffffu) ^ ((((c0 >> 23) ^ x0) & 1u) ? poly : 0u);
              c1 = ((c1 << 1) & 0xffffffu) ^ ((((c1 >> 23) ^ x1) & 1u) ? poly : 0u);
            }
            if (o16 && whole) { // parity aid: decision LLRs in natural order
              o16[(2 * pl) * long_sb + b * 8 + j]     = AR::out16(v.x);
              o16[(2 * pl + 1) * long_sb + b * 8 + j] = AR::out16(v.y);
            }
          }
        }
        if (!whole) {
          sbuf[(2 * pl) * sbs + b]     = (uint8_t)b0;
          sbuf[(2 * pl + 1) * sbs + b] = (uint8_t)b1;
        }
        if (whole && write && live) {
          if (wide) {
            w0 |= b0 << (8 * (b & 3));
            w1 |= b1 << (8 * (b & 3));
            if ((b & 3) == 3) {
              *reinterpret_cast<uint32_t*>(out + (2 * pl) * bps + (b & ~3u))     = w0;
              *reinterpret_cast<uint32_t*>(out + (2 * pl + 1) * bps + (b & ~3u)) = w1;
              w0 = w1 = 0;
            }
          } else {
            if ((2 * pl) * bps + b < out_bytes) {
              out[(2 * pl) * bps + b] = (uint8_t)b0;
            }
            if ((2 * pl + 1) * bps + b < out_bytes) {
              out[(2 * pl + 1) * bps + b] = (uint8_t)b1;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          t[j] = tn[j];
        }
      }
    }
    uint32_t crc = 0;
    if (crc_poly) {
      auto mulmod = [&](uint32_t a, uint32_t m) { // a(x) m(x) mod g(x), all below x^24
        uint32_t r = 0;
#pragma unroll 4
        for (int i = 23; i >= 0; i--) {
          r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? poly : 0u);
          r ^= ((m >> i) & 1u) ? a : 0u;
        }
        return r;
      };
      crc = mulmod(c0, p.crc_mult[2 * pl]) ^ mulmod(c1, p.crc_mult[2 * pl + 1]);
#pragma unroll
      for (int off = LPC / 2; off > 0; off >>= 1) {
        crc ^= __shfl_xor(crc, off, LPC);
      }
    }
    // ragged sub-blocks: with early stop the bytes are cut out only when the block has just passed its CRC or on the last
    // half iteration allowed (earlier attempts would be overwritten anyway)
    __syncthreads();
    if (!whole && write && (!crc_poly || crc == 0 || final_try)) {
      const uint32_t nbytes = K / 8;
      const uint32_t lim    = out_bytes < nbytes ? out_bytes : nbytes;
      const bool     al     = (reinterpret_cast<uintptr_t>(out) & 3) == 0;
      for (uint32_t w = pl; w * 4 < nbytes; w += LPC) {
        uint32_t d = (w * 32) / long_sb, k = w * 32 - d * long_sb;
        uint32_t word = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (w * 4 + t < nbytes) {
            // 8 bits of sub-block d from step k (zeros past its end), completed from the head of sub-block d + 1
            const uint8_t* q = sbuf + d * sbs + (k >> 3);
            uint32_t       v = ((((uint32_t)q[0] << 8) | q[1]) >> (8 - (k & 7))) & 0xffu;
            if (k + 8 > long_sb && d + 1 < (uint32_t)NB) {
              v |= (uint32_t)sbuf[(d + 1) * sbs] >> (long_sb - k);
            }
            word |= v << (8 * t);
          }
          k += 8;
          if (k >= long_sb) {
            k -= long_sb;
            d++;
          }
        }
        if (live) {
          if (al && w * 4 + 3 < lim) {
            *reinterpret_cast<uint32_t*>(out + 4 * w) = word;
          } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
              if (w * 4 + t < lim) {
                out[w * 4 + t] = (uint8_t)(word >> (8 * t));
              }
            }
          }
        }
      }
      if (o16) { // parity aid: decision LLRs in natural order
        const short* sd = reinterpret_cast<const short*>(D);
        for (uint32_t nn = pl; nn < K; nn += LPC) {
          const uint32_t d = nn / long_sb, k = nn % long_sb;
          const uint32_t e = (k * 64 + (lane / LPC) * LPC + (d >> 1)) * 2 + (d & 1);
          o16[nn] = AR::kIs8 ? (short)reinterpret_cast<const signed char*>(D)[e] : sd[e]; // (8-bit storage holds the plain int8 values)
        }
      }
    }
    __syncthreads();
    return crc;
  };
  bool     done = false; // early stop: the CRC of this code block has matched
  uint32_t noi  = 0;     // half iterations run for this code block in this launch (sch.c:424: cb_noi)

  for (uint32_t n = p.n_begin; n < p.n_end; n++) {
    const bool      dec1    = !(n & 1);
    const bool      has_app = dec1 && n > 0;
    const uint32_t* Y       = dec1 ? P0 : P1;
    const short*    xt      = dec1 ? TL : TL + 6;
    const short*    yt      = dec1 ? TL + 3 : TL + 9;

    // operands of 8 consecutive steps of this lane's two sub-blocks: x (systematic + a-priori), y (parity).
    // issue() only starts the loads (software pipelining: the next block is requested before the current
    // one is computed, so HBM latency hides behind a few hundred VALU instructions); prep() consumes them.
    struct Ops {
      uint32_t x[8], y[8], a[8];
    };
    auto issue = [&](uint32_t b, Ops& q) {
      if (dec1) {
        load_block_raw<AR::kIs8, NT>(S, (b & M_OWN) * 64 + lane, q.x);
      } else {
        issue_rows_raw<AR::kIs8, NT>(A2, b & M_ROW, lane, q.x);
      }
      load_block_raw<AR::kIs8, NT>(Y, (b & M_OWN) * 64 + lane, q.y);
      if (has_app) {
        issue_rows_raw<AR::kIs8, NT>(A1, b & M_ROW, lane, q.a);
      }
    };
    auto prep = [&](const Ops& q, s2(&xs)[8], s2(&ys)[8], s2(&ap)[8]) {
      uint32_t xr[8], ar[8], yr[8];
      if (dec1) {
        block_values<AR::kIs8>(q.x, xr);
      } else {
        rows_to_lane_v<AR::kIs8>(Tr, lane, q.x, xr);
      }
      if (has_app) {
        rows_to_lane_v<AR::kIs8>(Tr, lane, q.a, ar);
      }
      block_values<AR::kIs8>(q.y, yr);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        xs[j] = from_u(xr[j]);
        ys[j] = from_u(yr[j]);
        ap[j] = splat(0);
        if (has_app) {
          ap[j] = from_u(ar[j]);
          xs[j] = AR::add(ap[j], xs[j]);
        }
      }
    };
    Ops cur, nxt, nx2; // the two main passes keep TWO blocks in flight: HBM latency under load exceeds one block of compute

    s2 o[8];
    // ================= backward recursion (turbodecoder_win.h:551-681)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      o[i] = splat(-AR::kInf);
    }
    // pass 0: 40 steps on the head of every sub-block, all states unknown
    issue(TD_WIN_OVERLAP / 8 - 1, cur);
    for (int b = TD_WIN_OVERLAP / 8 - 1; b >= 0; b--) {
      issue(b > 0 ? b - 1 : nblk - 1, nxt); // after the warm-up, pass 1 starts at the last block
      s2 xs[8], ys[8], ap[8];
      prep(cur, xs, ys, ap);
#pragma unroll
      for (int j = 7; j >= 0; j--) {
        beta_step<AR>(o, xs[j], ys[j]);
        uint32_t k = b * 8 + j;
        if (AR::norm_at(k)) {
          AR::normalize(o);
        }
      }
      cur = nxt;
    }
    // hand every estimate to the previous sub-block; the last one starts from the tail trellis
    {
      short tr[8];
      tail_trellis<AR>(xt, yt, tr);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint32_t u   = to_u(o[i]);
        uint32_t nxt = __shfl_down(u, 1, LPC);
        uint32_t lo  = u >> 16;
        uint32_t hi  = (pl == LPC - 1) ? (uint32_t)(uint16_t)tr[i] : (nxt & 0xffffu);
        o[i]         = from_u(lo | (hi << 16));
      }
      uint32_t ck[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        ck[i] = to_u(o[i]);
      }
      store_block_v<AR::kIs8>(CK, (nblk & M_OWN) * 64 + lane, ck);
    }
    // pass 1: whole sub-block, keep a check-point at every block boundary
    if (nblk > 1) {
      issue(nblk - 2, nxt);
    }
    for (int b = (int)nblk - 1; b >= 0; b--) {
      if (b > 1) {
        issue(b - 2, nx2);
      }
      s2 xs[8], ys[8], ap[8];
      prep(cur, xs, ys, ap);
#pragma unroll
      for (int j = 7; j >= 0; j--) {
        uint32_t k = b * 8 + j;
        if (k < long_sb) {
          beta_step<AR>(o, xs[j], ys[j]);
          if (j == 0 && b > 0) {
            uint32_t ck[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
              ck[i] = to_u(o[i]);
            }
            store_block_v<AR::kIs8>(CK, ((uint32_t)b & M_OWN) * 64 + lane, ck);
          }
          if (AR::norm_at(k)) {
            AR::normalize(o);
          }
        }
      }
      cur = nxt;
      nxt = nx2;
    }
    __syncthreads(); // orders this lane's check-point stores before its loads below

    // ================= forward recursion + LLR (turbodecoder_win.h:684-832)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      o[i] = splat(-AR::kInf);
    }
    {
      const uint32_t w0 = long_sb - TD_WIN_OVERLAP;
      const uint32_t bl = (long_sb - 1) >> 3;
      issue(w0 >> 3, cur);
      for (uint32_t b = w0 >> 3; b <= bl; b++) {
        issue(b < bl ? b + 1 : 0, nxt); // after the warm-up the main pass starts at block 0
        s2 xs[8], ys[8], ap[8];
        prep(cur, xs, ys, ap);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          uint32_t k = b * 8 + j;
          if (k >= w0 && k < long_sb) {
            alpha_step<AR, false>(o, o, xs[j], ys[j]);
            uint32_t kk = k - w0;
            if (AR::norm_at(kk)) {
              AR::normalize(o);
            }
          }
        }
        cur = nxt;
      }
    }
    // hand every estimate to the next sub-block; the first one starts in state 0
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t u   = to_u(o[i]);
      uint32_t prv = __shfl_up(u, 1, LPC);
      uint32_t lo  = (pl == 0) ? (uint32_t)(uint16_t)(short)(i ? -AR::kInf : 0) : (prv >> 16);
      uint32_t hi  = u & 0xffffu;
      o[i]         = from_u(lo | (hi << 16));
    }

    const uint32_t* lut = dec1 ? p.deint : p.inter; // per (step, destination lane): row | source sub-blocks
    uint32_t*       dst = dec1 ? A2 : A1;
    // The subtractions of the NEXT half iteration (turbodecoder_iter.h:108,115) are applied on the way out:
    //   decoder 1: ext1 -= app1 (the a-priori it just used) before the interleaved copy goes to A2
    //   decoder 2: app1 = ext2 - ext1 ... and the ext1 value an output pairs with after de-interleaving is the very
    //              A2 value it had as systematic input at the same step and lane (app2[i] = ext1[inter[i]]), so the
    //              difference is formed before the permutation and no ext1 array is exchanged at all.
    // The raw SISO output is only needed by the hard decision: the LAST half iteration of a launch files it in D.
    const bool fuse = dec1 && n >= 2;
    const bool last = (n + 1 == p.n_end) || crc_poly; // with early stop every half iteration may be the last

    uint32_t ck[8], tr[8], ckn[8], trn[8];
    load_block_raw<AR::kIs8, NT>(CK, (1u & M_OWN) * 64 + lane, ck);
    load_lut(lut, pl, tr);
    if (nblk > 1) {
      issue(1, nxt);
    }
    for (uint32_t b = 0; b < nblk; b++) {
      const int len = (long_sb - b * 8) < 8 ? (int)(long_sb - b * 8) : 8;
      s2        xs[8], ys[8], ap[8];
      if (b + 2 < nblk) {
        issue(b + 2, nx2);
      }
      if (b + 1 < nblk) {
        load_block_raw<AR::kIs8, NT>(CK, ((b + 2) & M_OWN) * 64 + lane, ckn);
        load_lut(lut, (b + 1) * LPC + pl, trn);
      }
      prep(cur, xs, ys, ap);
      // re-derive beta[8b+1 .. 8b+len] (the stored, pre-normalisation values) from the check-point into
      // this lane's private LDS slots (registers are needed for the prefetched operands)
      {
        s2       st[8];
        uint32_t ckv[8];
        block_values<AR::kIs8>(ck, ckv);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          st[i] = from_u(ckv[i]);
        }
        Bl[len - 1][0][lane] = make_uint4(ckv[0], ckv[1], ckv[2], ckv[3]);
        Bl[len - 1][1][lane] = make_uint4(ckv[4], ckv[5], ckv[6], ckv[7]);
#pragma unroll
        for (int j = 6; j >= 0; j--) {
          if (j <= len - 2) {
            uint32_t idx = b * 8 + j + 2; // index of the stored value we start from
            if (idx != long_sb && AR::norm_at(idx)) {
              AR::normalize(st);
            }
            beta_step<AR>(st, xs[j + 1], ys[j + 1]);
            Bl[j][0][lane] = make_uint4(to_u(st[0]), to_u(st[1]), to_u(st[2]), to_u(st[3]));
            Bl[j][1][lane] = make_uint4(to_u(st[4]), to_u(st[5]), to_u(st[6]), to_u(st[7]));
          }
        }
      }
      uint32_t outv[8], rawv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        outv[j] = 0;
        rawv[j] = 0;
        if (j < len) {
          const uint4 b0 = Bl[j][0][lane], b1 = Bl[j][1][lane];
          const s2    B[8] = {from_u(b0.x), from_u(b0.y), from_u(b0.z), from_u(b0.w),
                              from_u(b1.x), from_u(b1.y), from_u(b1.z), from_u(b1.w)};
          s2       llr = alpha_step<AR, true>(o, B, xs[j], ys[j]);
          uint32_t k   = b * 8 + j;
          if (AR::norm_at(k)) {
            AR::normalize(o);
          }
          s2 proc = llr;
          if (fuse) {
            proc = AR::ex_sub(llr, ap[j], k == wrap_row);
          } else if (!dec1) {
            proc = AR::ex_sub(llr, xs[j], (tr[j] & 0xffffu) == wrap_row); // wrap flag: row of the DESTINATION element
          }
          outv[j] = to_u(proc);
          rawv[j] = to_u(llr);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (j < len) {
          const uint32_t row = tr[j] & 0xffffu & M_OWN;
          store_row<AR::kIs8>(dst, row, lane, permute_pair<LPC>(outv[j], tr[j] >> 16));
          if (last) {
            // what tdec_decision_byte reads (turbodecoder.c:370-378), in natural order: ext1 after decoder 1,
            // the de-interleaved ext2 (= app1 before the subtraction) after decoder 2
            if (dec1) {
              store_row<AR::kIs8>(D, (b * 8 + j) & M_OWN, lane, rawv[j]);
            } else {
              store_row<AR::kIs8>(D, row, lane, permute_pair<LPC>(rawv[j], tr[j] >> 16));
            }
          }
        }
      }
      cur = nxt;
      nxt = nx2;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        ck[i] = ckn[i];
        tr[i] = trn[i];
      }
    }
    __syncthreads();
    if (crc_poly) {
      // decode_tb_cb (sch.c:420-454): hard bits + CRC after every half iteration; a code block stops at its first
      // match (its bits are written then and never again), the wave stops when all its code blocks have
      const bool     fin = n + 1 == p.n_end;
      const bool     wr  = !done; // bits of the last half iteration THIS code block took part in
      const uint32_t crc = decide(wr, fin);
      if (!done) {
        noi++;
        done = crc == 0;
      }
      set_masks(done || !live);
      __syncthreads();
      if (__all(done || !live) || fin) {
        break;
      }
    }
  }

#undef M_OWN
#undef M_ROW
  if (!crc_poly) {
    decide(true, true);
  } else if (p.noi && live && pl == 0) {
    p.noi[cb_raw]    = (int)noi;
    p.crc_ok[cb_raw] = done ? 1 : 0;
  }
}

// The product kernel: one workgroup (= one wave) per unit, two waves per SIMD.
template <int LPC, class AR, bool ES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void tdec_win_kernel(const WinParams p)
{
  __shared__ uint4    Bl[8][2][64];
  __shared__ uint32_t Tr[512];
  tdec_win_unit<LPC, AR, ES>(p, blockIdx.x, blockIdx.x, threadIdx.x, Bl, Tr);
}

#ifdef SRSRAN_HIP_WITH_VARIANTS // compiled into tools/probe/lib/libsrsran_phy_hip_variants.so only (srslte_amd/build.py --variants)
// ---- measured alternatives of the launch shape (profiles/r02_turbo_variants.txt, DESIGN.md par. 3.2); selected with
// SRSRAN_HIP_TDEC_VARIANT for the 16-sub-block int16 decoder without early stop only, never by default.
// "waves1": one wave per SIMD (half the code blocks in flight, up to 512 VGPRs per lane).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void tdec_win_kernel_waves1(const WinParams p)
{
  __shared__ uint4    Bl[8][2][64];
  __shared__ uint32_t Tr[512];
  tdec_win_unit<8, Ar16, false>(p, blockIdx.x, blockIdx.x, threadIdx.x, Bl, Tr);
}
// "persistent": the grid holds only as many workgroups as the chip keeps resident (p.max_resident), every workgroup owns ONE slab and
// takes units from a counter until the batch is used up, so the workspace could be sized by the residency (1.2 GB) instead of the
// batch (5.6 GB for 65,520 blocks).  The parameter block and the lane index go through an opaque asm so that what derives from
// them is not hoisted in front of the loop (without that: 333 spilled VGPRs; with it 47, all in the input extraction).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void tdec_win_kernel_persistent(const WinParams p)
{
  __shared__ uint4    Bl[8][2][64];
  __shared__ uint32_t Tr[512];
  uint32_t wb = blockIdx.x;
  while (wb < p.n_units) {
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const WinParams& q = *(const WinParams*)ka;
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    tdec_win_unit<8, Ar16, false>(q, wb, blockIdx.x, lane, Bl, Tr);
    __syncthreads();
    uint32_t nx = 0;
    if (threadIdx.x == 0) {
      nx = gridDim.x + atomicAdd(q.unit_counter, 1u);
    }
    wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
  }
}

#endif // SRSRAN_HIP_WITH_VARIANTS
// ------------------------------------------------------------------------------------------------
// Scalar decoder (turbodecoder_gen.c): one lane per code block, wrapping int16, beta kept in HBM.
// Used for K <= 400 (AUTO) or SRSRAN_TDEC_GENERIC.  Vectors are stored lane-interleaved
// [index][64 lanes] so that a wave's accesses coalesce.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gen_acs_beta(short (&old)[8], short x, short y)
{
  short xy = wrap16(x + y);
  short m_b[8], nw[8];
  m_b[0] = wrap16(old[4] + xy);
  m_b[1] = old[4];
  m_b[2] = wrap16(old[5] + y);
  m_b[3] = wrap16(old[5] + x);
  m_b[4] = wrap16(old[6] + x);
  m_b[5] = wrap16(old[6] + y);
  m_b[6] = old[7];
  m_b[7] = wrap16(old[7] + xy);
  nw[0] = old[0];
  nw[1] = wrap16(old[0] + xy);
  nw[2] = wrap16(old[1] + x);
  nw[3] = wrap16(old[1] + y);
  nw[4] = wrap16(old[2] + y);
  nw[5] = wrap16(old[2] + x);
  nw[6] = wrap16(old[3] + xy);
  nw[7] = old[3];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    old[i] = m_b[i] > nw[i] ? m_b[i] : nw[i];
  }
}

__global__ __launch_bounds__(64) void tdec_gen_kernel(const GenParams p)
{
  const int lane = threadIdx.x;
  const int cb   = blockIdx.x * 64 + lane;
  if (cb >= p.n_cb) {
    return;
  }
  const uint32_t K  = p.K;
  const uint32_t L  = K + 4; // K + 3 tail (+1 for the beta terminal state)
  // per-wave slab, element (array, index, lane): ((array_base + index) * 64 + lane)
  short* ws = p.ws + (size_t)blockIdx.x * p.ws_stride;
#define GV(base, idx) ws[((size_t)(base) + (idx)) * 64 + lane]
  const uint32_t oS = 0, oP0 = L, oP1 = 2 * L, oA1 = 3 * L, oA2 = 4 * L, oE1 = 5 * L, oE2 = 6 * L, oB = 7 * L;
  // beta: 8 * (K+4) from oB

  if (p.n_begin == 0) {
    // int8 input: the 8-bit API widens to int16 when no 8-bit decoder takes this K (turbodecoder.c:455-478)
    const size_t       in_off = p.desc ? (size_t)p.desc[cb].in_off : (size_t)cb * p.in_stride;
    const short*       in16 = p.input + in_off;
    const signed char* in8  = reinterpret_cast<const signed char*>(p.input) + in_off;
    auto               in   = [&](uint32_t i) -> short { return p.in_is8 ? (short)in8[i] : in16[i]; };
    for (uint32_t i = 0; i < K; i++) { // turbodecoder_gen.c:238-258
      GV(oS, i)  = in(3 * i);
      GV(oP0, i) = in(3 * i + 1);
      GV(oP1, i) = in(3 * i + 2);
    }
    for (uint32_t i = K; i < K + 3; i++) {
      GV(oS, i)  = in(3 * K + 2 * (i - K));
      GV(oP0, i) = in(3 * K + 2 * (i - K) + 1);
      GV(oA2, i) = in(3 * K + 6 + 2 * (i - K));
      GV(oP1, i) = in(3 * K + 6 + 2 * (i - K) + 1);
    }
  }
  const uint16_t* inter   = p.inter;
  const uint16_t* deinter = p.deinter;
  uint32_t        n_run    = p.n_end; // half iterations completed when the loop is left
  bool            crc_good = false;

  for (uint32_t n = p.n_begin; n < p.n_end; n++) {
    const bool     dec1    = !(n & 1);
    const bool     has_app = dec1 && n > 0;
    const uint32_t oX = dec1 ? oS : oA2, oY = dec1 ? oP0 : oP1, oOut = dec1 ? oE1 : oE2;
    if (dec1) {
      if (n) {
        for (uint32_t i = 0; i < K; i++) {
          GV(oA1, i) = wrap16(GV(oA1, i) - GV(oE1, i));
        }
      }
    } else {
      for (uint32_t i = 0; i < K; i++) {
        short e = GV(oE1, i);
        if (n > 1) {
          e          = wrap16(e - GV(oA1, i));
          GV(oE1, i) = e;
        }
        GV(oA2, deinter[i]) = e;
      }
    }
    // map_gen_beta (turbodecoder_gen.c:58-112)
    short old[8];
    old[0] = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) {
      old[i] = -TD_INF;
    }
    const uint32_t end = K + 3;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      GV(oB + 8 * end, i) = old[i];
    }
    for (int k = (int)end - 1; k >= 0; k--) {
      short x = GV(oX, k);
      if (has_app && (uint32_t)k < K) {
        x = wrap16(x + GV(oA1, k));
      }
      short y = GV(oY, k);
      gen_acs_beta(old, x, y);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        GV(oB + 8 * k, i) = old[i];
      }
      if ((k % 4) == 0 && (uint32_t)k < K) {
#pragma unroll
        for (int i = 1; i < 8; i++) {
          old[i] = wrap16(old[i] - old[0]);
        }
        old[0] = 0;
      }
    }
    // map_gen_alpha (turbodecoder_gen.c:114-198)
    old[0] = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) {
      old[i] = -TD_INF;
    }
    for (uint32_t k = 1; k < K + 1; k++) {
      short x = GV(oX, k - 1);
      if (has_app) {
        x = wrap16(x + GV(oA1, k - 1));
      }
      short y  = GV(oY, k - 1);
      short xy = wrap16(x + y);
      short m_b[8], nw[8];
      m_b[0] = old[0];
      m_b[1] = wrap16(old[3] + y);
      m_b[2] = wrap16(old[4] + y);
      m_b[3] = old[7];
      m_b[4] = old[1];
      m_b[5] = wrap16(old[2] + y);
      m_b[6] = wrap16(old[5] + y);
      m_b[7] = old[6];
      nw[0] = wrap16(old[1] + xy);
      nw[1] = wrap16(old[2] + x);
      nw[2] = wrap16(old[5] + x);
      nw[3] = wrap16(old[6] + xy);
      nw[4] = wrap16(old[0] + xy);
      nw[5] = wrap16(old[3] + x);
      nw[6] = wrap16(old[4] + x);
      nw[7] = wrap16(old[7] + xy);
      short m1 = 0, m0 = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        short bq = GV(oB + 8 * k, i);
        short v0 = wrap16(m_b[i] + bq);
        short v1 = wrap16(nw[i] + bq);
        m0 = (i == 0) ? v0 : (v0 > m0 ? v0 : m0);
        m1 = (i == 0) ? v1 : (v1 > m1 ? v1 : m1);
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        old[i] = m_b[i] > nw[i] ? m_b[i] : nw[i];
      }
      if ((k % 4) == 0) {
#pragma unroll
        for (int i = 1; i < 8; i++) {
          old[i] = wrap16(old[i] - old[0]);
        }
        old[0] = 0;
      }
      GV(oOut, k - 1) = wrap16(m1 - m0);
    }
    if (!dec1) {
      for (uint32_t i = 0; i < K; i++) {
        GV(oA1, inter[i]) = GV(oE2, i);
      }
    }
    n_run = n + 1;
    if (p.crc_poly) {
      // decode_tb_cb (sch.c:420-454): the checksum of the K hard bits after every half iteration (crc.c:92-140: MSB first,
      // zero initial state; zero = the block is good); this lane's block stops at its first match
      const uint32_t oC = (n_run & 1) ? oE1 : oA1;
      const uint32_t g  = p.crc_poly & 0xffffffu;
      uint32_t       c  = 0;
      for (uint32_t i = 0; i < K; i++) {
        const uint32_t x = GV(oC, i) > 0 ? 1u : 0u;
        c = ((c << 1) & 0xffffffu) ^ ((((c >> 23) ^ x) & 1u) ? g : 0u);
      }
      if (c == 0) {
        crc_good = true;
        break;
      }
    }
  }
  if (p.noi) {
    p.noi[cb] = (int)(n_run - p.n_begin);
  }
  if (p.crc_ok) {
    p.crc_ok[cb] = crc_good ? 1 : 0;
  }
  // decision (turbodecoder.c:370-378, turbodecoder_gen.c:260-277)
  const uint32_t oD        = (n_run & 1) ? oE1 : oA1;
  uint8_t*       out       = p.output + (p.desc ? (size_t)p.desc[cb].out_off : (size_t)cb * p.out_stride);
  const uint32_t out_bytes = p.desc ? p.desc[cb].out_bytes : K / 8;
  for (uint32_t jb = 0; jb < out_bytes; jb++) {
    uint32_t byte = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      byte |= (GV(oD, jb * 8 + t) > 0 ? 0x80u : 0u) >> t;
    }
    out[jb] = (uint8_t)byte;
  }
  if (p.dec_llr) {
    short* o16 = p.dec_llr + (size_t)cb * K;
    for (uint32_t i = 0; i < K; i++) {
      o16[i] = GV(oD, i);
    }
  }
#undef GV
}


// ------------------------------------------------------------------------------------------------ launchers

template <bool ES>
static hipError_t launch_win_es(int nb, bool arith8, const WinParams& p, hipStream_t stream)
{
  const int lpc = nb / 2;
  dim3      grid(ceil_div(p.n_cb, 64 / lpc));
#ifdef SRSRAN_HIP_WITH_VARIANTS
  if (!ES && !arith8 && nb == 16 && p.variant == 1) {
    hipLaunchKernelGGL(tdec_win_kernel_waves1, grid, dim3(64), 0, stream, p);
  } else if (!ES && !arith8 && nb == 16 && p.variant == 2 && p.unit_counter) {
    hipLaunchKernelGGL(tdec_win_kernel_persistent, dim3(p.n_units < p.max_resident ? p.n_units : p.max_resident), dim3(64), 0, stream, p);
  } else
#endif
  if (!arith8 && nb == 16) {
    hipLaunchKernelGGL((tdec_win_kernel<8, Ar16, ES>), grid, dim3(64), 0, stream, p);
  } else if (!arith8 && nb == 8) {
    hipLaunchKernelGGL((tdec_win_kernel<4, Ar16, ES>), grid, dim3(64), 0, stream, p);
  } else if (arith8 && nb == 16) {
    hipLaunchKernelGGL((tdec_win_kernel<8, Ar8, ES>), grid, dim3(64), 0, stream, p);
  } else if (arith8 && nb == 32) {
    hipLaunchKernelGGL((tdec_win_kernel<16, Ar8, ES>), grid, dim3(64), 0, stream, p);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_win(int nb, bool arith8, const WinParams& p, hipStream_t stream)
{
  return (p.crc_poly || p.desc) ? launch_win_es<true>(nb, arith8, p, stream) : launch_win_es<false>(nb, arith8, p, stream);
}

hipError_t launch_gen(const GenParams& p, hipStream_t stream)
{
  dim3 grid(ceil_div(p.n_cb, 64));
  hipLaunchKernelGGL(tdec_gen_kernel, grid, dim3(64), 0, stream, p);
  return hipGetLastError();
}

uint32_t win_elem_index(int nb, uint32_t k, uint32_t d)
{
  return nb == 32 ? elem_index<16>(k, d) : (nb == 16 ? elem_index<8>(k, d) : elem_index<4>(k, d));
}

} // namespace turbo
} // namespace phyhip
