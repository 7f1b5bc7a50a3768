// turbo_lat_kernels.hip -- LATENCY kernel of the LTE turbo decoder for gfx950: one code block per wave, trellis STATES across lanes.
//
// Same decoders, bit for bit, as turbo_kernels.hip (the reference's window decoders: turbodecoder_win.h, turbodecoder_iter.h), another
// mapping.  The throughput kernel gives a lane two sub-blocks and all 8 states (8 code blocks per wave): every trellis step is ~130 packed
// instructions that ONE wave issues for its 8 blocks -- right when tens of thousands of blocks are in flight, 250 us per half iteration when
// the batch is one subframe's worth (the reference decodes one subframe per worker call: cc_worker.cc:212-231, sch.c:389-492) and most of
// the chip idles.  Here a code block owns 8 lanes per sub-block pair:
//     lane = 8 * pair + state-slot        (16 sub-blocks: 64 lanes = one wave per block; 8 sub-blocks: 32 lanes, two blocks per wave)
//   * each lane holds ONE state metric (int16x2: the two sub-blocks of its pair), so a step is ~60 instructions per wave for the block:
//     the add-compare-select is two saturating adds and a max against the PARTNER lane's metric;
//   * the trellis is a shift register (new state = (u, b2, b1) from (b2, b1, b0)), so the butterflies are done in place and the labelling
//     of the 8 slots rotates with the step index: slot bits (x2 x1 x0) hold state (x2 x1 x0) at steps = 0 mod 3, (x0 x2 x1) at 1 mod 3,
//     (x1 x0 x2) at 2 mod 3; the partner is slot ^ 1, ^ 2, ^ 4 (DPP quad_perm / row shifts), never LDS;
//   * alpha, beta and the LLR of a step use the SAME two branch metrics per lane (the code's butterflies are symmetric), selected once per
//     step from {0, x, y, x + y} by two lane masks that are compile-time constants per residue;
//   * max-log-MAP output = two 8-lane max reductions (DPP), normalisation = a broadcast of slot 0 / an 8-lane max (8-bit arithmetic);
//   * operands are fetched straight into registers by broadcast loads (the 8 lanes of a pair read the same 32 bytes), two 8-step blocks
//     ahead; backward metrics are check-pointed every 8 steps and re-derived in registers, as in the throughput kernel;
//   * the QPP exchange of an 8-step block is two ds_bpermute for all its 8 rows at once.
// Everything that defines results -- saturation, normalisation schedule, 40-step warm-ups, tail trellis, extrinsic subtraction order,
// decision source, CRC early stop -- is the throughput kernel's (shared arithmetic policies: turbo_arith.h); tests/test_gpu_turbo.py and
// tests/test_gpu_sch.py run both kernels against the oracle.  Selected by batch size in turbo_host.cpp (or SRSRAN_HIP_TDEC_LAT).
#include "hip_common.h"
#include "turbo_arith.h"
#include "turbo_device.h"
#include "turbo_lat_common.h"

#include <type_traits>

namespace phyhip {
namespace turbo {
namespace lat {


// dwords of workspace per code block
__host__ __device__ inline uint32_t ws_dwords(uint32_t K, int nb)
{
  const uint32_t lpc = nb / 2, long_sb = K / nb, nblk = (long_sb + 7) / 8;
  return 6 * nblk * lpc * 8 + (nblk + 1) * lpc * 8 + 8;
}

// DUAL (16 sub-blocks, 16-bit): TWO waves per code block with the same lane mapping -- wave 0 runs the forward recursion, wave 1 the backward one, at
// the same time; each files the metrics of the first half of its way in LDS and forms the outputs of the second half against what the other one filed.
template <int LPC, class AR, bool ES, bool DUAL = false>
__global__ __launch_bounds__(DUAL ? 128 : (LPC > 8 ? 8 * LPC : 64)) void tdec_lat_kernel(const WinParams p)
{
  static_assert(!DUAL || LPC == 8, "the two-wave form exists for the 16-sub-block decoders");
  constexpr int NB  = 2 * LPC;
  constexpr int G   = 8 * LPC;            // lanes per code block: 32 (two blocks per wave), 64 (one wave) or 128 (two waves, 32 sub-blocks)
  constexpr int BPW = G >= 64 ? 1 : 64 / G; // code blocks per workgroup
  __shared__ uint32_t simg_all[BPW][NB * ((6144 / NB / 8 + 1 + 3) / 4)]; // hard bits of the code block(s): image of the decision
  __shared__ uint32_t xch[2][G > 64 ? G : 1];                  // lane exchange across the two waves of a 128-lane block
  __shared__ uint32_t xcrc[2];

  const int      lane = DUAL ? (int)(threadIdx.x & 63u) : (int)threadIdx.x; // (DUAL: both waves number their lanes alike)
  const int      wave = (int)(threadIdx.x >> 6);
  const int      grp  = lane / G, li = lane % G, pl = li >> 3, slot = li & 7;
  const int      gbase = grp * G;
  const int      cb_raw = (int)blockIdx.x * BPW + grp;
  const bool     live = cb_raw < p.n_cb;
  const int      cb   = live ? cb_raw : p.n_cb - 1;
  const uint32_t K = p.K, long_sb = K / NB, nblk = (long_sb + 7) >> 3;
  const uint32_t AW = nblk * LPC * 8;
  const uint32_t crc_poly = ES ? p.crc_poly : 0u;
  const CbDesc*  desc     = ES ? p.desc : nullptr;
  const LaneK    lk       = lane_consts(slot);

  uint32_t* ws = p.ws + (size_t)cb_raw * p.ws_stride; // dead groups own a slot too (they decode a copy of the last block, write no output)
  uint32_t* S  = ws;
  uint32_t* P0 = ws + AW;
  uint32_t* P1 = ws + 2 * AW;
  uint32_t* A1 = ws + 3 * AW;
  uint32_t* D  = ws + 4 * AW;
  uint32_t* A2 = ws + 5 * AW;
  uint32_t* CK = ws + 6 * AW;                                       // (nblk + 1) check-points of G dwords
  short*    TL = reinterpret_cast<short*>(CK + (size_t)(nblk + 1) * G); // 12 tail LLRs

  // two values of `v` held by other lanes of the code block (absolute lane numbers): ds_bpermute inside a wave, an LDS image + one barrier
  // when the block spans two waves (`par` alternates the image so that no second barrier is needed before the next use)
  int  xpar  = 0;
  auto read2 = [&](uint32_t v, int la, int lc, uint32_t& a, uint32_t& c) {
    if constexpr (G <= 64) {
      a = (uint32_t)__shfl((int)v, la, 64);
      c = (uint32_t)__shfl((int)v, lc, 64);
    } else {
      xch[xpar][lane] = v;
      __syncthreads();
      a = xch[xpar][la];
      c = xch[xpar][lc];
      xpar ^= 1;
    }
  };

  // ---- phase 0: input extraction (turbodecoder_win.h:888-930 / turbodecoder_iter.h:58-70,88-102): element (step 8 b + j, pair pl) per lane
  if (p.n_begin == 0) {
    const size_t in_off = desc ? (size_t)desc[cb].in_off : (size_t)cb * p.in_stride;
    auto         fetch  = [&](uint32_t e) -> short {
      return p.in_is8 ? AR::conv_in((int)reinterpret_cast<const signed char*>(p.input)[in_off + e]) : AR::conv_in((int)p.input[in_off + e]);
    };
    if (p.sb_layout) {
      // rm_turbo layout (what the de-matcher leaves in the soft buffer): element (step k, sub-block d) of stream a at a (K + 32) + k NB + d, so the
      // two sub-blocks of a pair are ONE aligned 32-bit (16-bit soft bits) or 16-bit (8-bit soft bits) word, a wave reads 8 steps x NB elements
      // = one contiguous run per stream, and four 8-step blocks are requested before the first is stored (the element-wise loop below: two
      // dependent 2-byte loads per stream and block, 8 us more per launch)
      const uint8_t* in8 = reinterpret_cast<const uint8_t*>(p.input) + (p.in_is8 ? in_off : 2 * in_off);
      auto           pair_word = [&](uint32_t e) -> uint32_t { // elements e, e + 1 (e even) through AR::conv_in, as `fetch` does
        if (p.in_is8) {
          const uint32_t w = *reinterpret_cast<const uint16_t*>(in8 + e);
          return (uint32_t)(uint16_t)AR::conv_in((int)(signed char)(w & 0xffu)) | ((uint32_t)(uint16_t)AR::conv_in((int)(signed char)(w >> 8)) << 16);
        }
        const uint32_t w = *reinterpret_cast<const uint32_t*>(in8 + 2 * (size_t)e);
        return (uint32_t)(uint16_t)AR::conv_in((int)(short)(w & 0xffffu)) | ((uint32_t)(uint16_t)AR::conv_in((int)(short)(w >> 16)) << 16);
      };
      constexpr uint32_t UB = 4;
      for (uint32_t b0 = DUAL ? UB * (uint32_t)wave : 0u; b0 < nblk; b0 += DUAL ? 2 * UB : UB) { // (DUAL: the two waves share the blocks)
        uint32_t v[UB][3];
#pragma unroll
        for (uint32_t u = 0; u < UB; u++) {
          const uint32_t k  = (b0 + u) * 8 + slot;
          const uint32_t kk = k < long_sb ? k : long_sb - 1;
#pragma unroll
          for (int a3 = 0; a3 < 3; a3++) {
            v[u][a3] = pair_word(a3 * (K + 32) + kk * NB + 2 * pl);
          }
        }
#pragma unroll
        for (uint32_t u = 0; u < UB; u++) {
          if (b0 + u < nblk) {
            const uint32_t at = ((b0 + u) * LPC + pl) * 8 + slot;
            S[at]  = v[u][0];
            P0[at] = v[u][1];
            P1[at] = v[u][2];
          }
        }
      }
    }
    for (uint32_t b = (p.sb_layout ? nblk : 0) + (DUAL ? (uint32_t)wave : 0u); b < nblk; b += DUAL ? 2 : 1) {
      const uint32_t k  = b * 8 + slot;
      const uint32_t kk = k < long_sb ? k : long_sb - 1;
      uint32_t       v[3];
#pragma unroll
      for (int a3 = 0; a3 < 3; a3++) {
        short lo, hi;
        if (p.sb_layout) { // (taken above)
          lo = fetch(a3 * (K + 32) + kk * NB + 2 * pl);
          hi = fetch(a3 * (K + 32) + kk * NB + 2 * pl + 1);
        } else { // natural order [s p0 p1] x K
          lo = fetch(3 * ((2 * pl) * long_sb + kk) + a3);
          hi = fetch(3 * ((2 * pl + 1) * long_sb + kk) + a3);
        }
        v[a3] = (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16);
      }
      const uint32_t at = (b * LPC + pl) * 8 + slot;
      S[at]  = v[0];
      P0[at] = v[1];
      P1[at] = v[2];
    }
    if (li < 12) {
      const uint32_t tb = p.sb_layout ? 3 * (K + 32) : 3 * K;
      // TL: [0..2] syst tail, [3..5] parity0 tail, [6..8] app2 tail, [9..11] parity1 tail
      const int grp4 = li / 3, i3 = li % 3;
      const uint32_t e = grp4 == 0 ? tb + 2 * i3 : (grp4 == 1 ? tb + 2 * i3 + 1 : (grp4 == 2 ? tb + 6 + 2 * i3 : tb + 6 + 2 * i3 + 1));
      TL[li] = fetch(e);
    }
  }
  const uint32_t wrap_row = (AR::kIs8 && (K & 31u)) ? long_sb - 1 : 0xffffffffu;
  __syncthreads();

  uint8_t*       out       = p.output + (desc ? (size_t)desc[cb].out_off : (size_t)cb * p.out_stride);
  const uint32_t out_bytes = desc ? desc[cb].out_bytes : K / 8;
  const int      rW        = (int)(long_sb % 3); // residue of the step index one past a sub-block

  // ---- hard decision (turbodecoder.c:370-378, turbodecoder_win.h:973-993) + CRC of the K bits (sch.c:430-447).
  // A pass that may be the last files its raw outputs in D in natural order (one store per lane and 8-step block, next to the exchange store); the
  // decision then loads D eight blocks at a time and turns signs into image bytes (sub-block d at simg + d * sbs4 bytes, MSB first, zero padded)
  // with one wave-wide ballot per sub-block of a pair: the 8 lanes of a pair hold the 8 steps of a block = one byte.  History: walking D one block
  // per memory round trip cost a third of an early-stop half iteration; collecting the bits inside the block loop (LDS atomics, byte stores into a
  // scratch, two 64-bit registers per lane) cost 11.5 of its 69.5 us whichever way; this form costs the same per half iteration (the store 6.7 us,
  // the pass over D 5 us: tools/measure/es_time.py against builds without either) but needs no image clearing and 25 registers fewer (72 instead of
  // 97 AGPRs of spilled state), which shows in the launch's fixed part: 832 blocks 0.182 -> 0.158 ms at one half iteration.
  uint8_t*       simg   = reinterpret_cast<uint8_t*>(&simg_all[grp][0]);
  const uint32_t sbs4   = (nblk + 1 + 3) & ~3u; // bytes per sub-block in the image
  for (uint32_t w = li; w < (uint32_t)NB * (sbs4 >> 2); w += G) {
    simg_all[grp][w] = 0; // (bytes behind a sub-block's last block stay zero)
  }
  auto           decide = [&](bool write, bool final_try) -> uint32_t {
    const bool     whole = (long_sb & 7) == 0;
    const uint32_t bps   = long_sb >> 3;
    const uint32_t poly  = crc_poly & 0xffffffu;
    uint32_t       crc   = 0;
    {
      const int sh = (lane & 63) & ~7; // this pair's byte in the ballots of its wave
      constexpr uint32_t UD = 8; // loads in flight (24 were slower: the registers they take are spilled state of the caller)
      for (uint32_t b0 = DUAL ? UD * (uint32_t)wave : 0u; b0 < nblk; b0 += DUAL ? 2 * UD : UD) { // (DUAL: the two waves share the blocks, the words and the bytes)
        uint32_t r[UD];
#pragma unroll
        for (uint32_t u = 0; u < UD; u++) {
          const uint32_t b = b0 + u < nblk ? b0 + u : nblk - 1;
          r[u]             = D[b * G + li];
        }
#pragma unroll
        for (uint32_t u = 0; u < UD; u++) {
          const uint32_t b = b0 + u;
          if (b < nblk) { // (uniform)
            const bool               in = b * 8 + slot < long_sb;
            const s2                 v  = from_u(r[u]);
            const unsigned long long mx = __ballot(in && v.x > 0), my = __ballot(in && v.y > 0);
            if (slot == 0) {
              simg[(2 * pl) * sbs4 + b]     = (uint8_t)(__brev((uint32_t)(mx >> sh) & 0xffu) >> 24);
              simg[(2 * pl + 1) * sbs4 + b] = (uint8_t)(__brev((uint32_t)(my >> sh) & 0xffu) >> 24);
            }
          }
        }
      }
      __syncthreads();
    }
    if (crc_poly) {
      // the CRC is linear: every lane runs the bit-serial register (crc.c:92-140, zero start) over whole 32-bit words of the image and shifts its
      // remainder into place with x^(bits behind the word) mod g (p.crc_mult: one multiplier per word of the image, turbo_host.cpp)
      auto mulmod = [&](uint32_t a, uint32_t m) { // a(x) m(x) mod g(x), all below x^24
        uint32_t r = 0;
#pragma unroll 4
        for (int i = 23; i >= 0; i--) {
          r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? poly : 0u);
          r ^= ((m >> i) & 1u) ? a : 0u;
        }
        return r;
      };
      const uint32_t wps = sbs4 >> 2, nw = (uint32_t)NB * wps;
      for (uint32_t w = (uint32_t)li + (DUAL ? 64u * (uint32_t)wave : 0u); w < nw; w += DUAL ? 2 * G : G) {
        const uint32_t d = w / wps, wi = w - d * wps;
        const int      nbit = (int)long_sb - (int)(32 * wi) > 32 ? 32 : (int)long_sb - (int)(32 * wi);
        if (nbit > 0) {
          const uint32_t word = reinterpret_cast<const uint32_t*>(simg)[w];
          uint32_t       c    = 0;
          for (int t = 0; t < nbit; t++) {
            const uint32_t x = (word >> (8 * (t >> 3) + 7 - (t & 7))) & 1u; // byte t / 8 of the word, MSB first
            c = ((c << 1) & 0xffffffu) ^ ((((c >> 23) ^ x) & 1u) ? poly : 0u);
          }
          crc ^= mulmod(c, p.crc_mult[w]);
        }
      }
#pragma unroll
      for (int off = (G < 64 ? G : 64) / 2; off > 0; off >>= 1) {
        crc ^= __shfl_xor(crc, off, G < 64 ? G : 64);
      }
      if constexpr (G > 64 || DUAL) { // the two waves of the block
        if ((lane & 63) == 0) {
          xcrc[wave] = crc;
        }
        __syncthreads();
        crc = xcrc[0] ^ xcrc[1];
        __syncthreads();
      }
    }
    if (write && live && (!crc_poly || crc == 0 || final_try)) {
      const uint32_t nbytes = K / 8;
      const uint32_t lim    = out_bytes < nbytes ? out_bytes : nbytes;
      for (uint32_t w = (uint32_t)li + (DUAL ? 64u * (uint32_t)wave : 0u); w < nbytes; w += DUAL ? 2 * G : G) {
        uint32_t v;
        if (whole) {
          const uint32_t d = w / bps;
          v                = simg[d * sbs4 + (w - d * bps)];
        } else {
          // ragged sub-blocks: 8 bits of sub-block d from step k (zeros past its end), completed from the head of sub-block d + 1
          const uint32_t d = (w * 8) / long_sb, k = w * 8 - d * long_sb;
          const uint8_t* q = simg + d * sbs4 + (k >> 3);
          v                = ((((uint32_t)q[0] << 8) | q[1]) >> (8 - (k & 7))) & 0xffu;
          if (k + 8 > long_sb && d + 1 < (uint32_t)NB) {
            v |= (uint32_t)simg[(d + 1) * sbs4] >> (long_sb - k);
          }
        }
        if (w < lim) {
          out[w] = (uint8_t)v;
        }
      }
    }
    if (p.dec_llr && live && write) { // parity aid: decision LLRs in natural order, from D
      short* o16 = p.dec_llr + (size_t)cb * K;
      for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t k = b * 8 + slot;
        if (k < long_sb) {
          const s2 v = from_u(D[(b * LPC + pl) * 8 + slot]);
          o16[(2 * pl) * long_sb + k]     = AR::out16(v.x);
          o16[(2 * pl + 1) * long_sb + k] = AR::out16(v.y);
        }
      }
    }
    __syncthreads();
    return crc;
  };

  bool     done = false;
  uint32_t noi  = 0;

  // one half iteration (turbodecoder_iter.h:72-141); DEC1 is a template parameter so that nothing inside the step loops branches on it
  auto half_iteration = [&](auto DEC1C, const uint32_t n) {
    constexpr bool  dec1    = decltype(DEC1C)::value;
    const bool      has_app = dec1 && n > 0;
    const uint32_t* X       = dec1 ? S : A2;
    const uint32_t* Y       = dec1 ? P0 : P1;
    const short*    xt      = dec1 ? TL : TL + 6;
    const short*    yt      = dec1 ? TL + 3 : TL + 9;

    // Operands of an 8-step block live in one of THREE register sets; the block loops are unrolled by three with the sets in fixed roles
    // (set = block index mod 3), so a set is filled two blocks before it is used and never copied: a copy of registers that a load is
    // still filling would wait for the load (the first version of this kernel did that and spent 40 % of its time in s_waitcnt vmcnt(0)).
    // The labelling residue of a block's first step is (8 b) mod 3 = (2 b) mod 3: a compile-time constant per set as well.
    // (held as the packed type they are used as: sets written as 32-bit integers and read back as int16 pairs stay in SCRATCH memory -- the compiler folds
    // the bit cast into the load, and a stack slot that is stored as one type and loaded as another is not promoted to registers: every operand went
    // global load -> wait -> scratch store -> scratch load, 30 scratch operations per 8-step block, which is what "two blocks ahead" never overlapped)
    struct Ops {
      s2 x[8], y[8], a[8];
    };
    auto load8s = [&](const uint32_t* q, s2(&r)[8]) {
      const uint4 a = *reinterpret_cast<const uint4*>(q), c = *reinterpret_cast<const uint4*>(q + 4);
      r[0] = from_u(a.x), r[1] = from_u(a.y), r[2] = from_u(a.z), r[3] = from_u(a.w), r[4] = from_u(c.x), r[5] = from_u(c.y), r[6] = from_u(c.z), r[7] = from_u(c.w);
    };
    Ops  buf0, buf1, buf2; // (three variables, not an array: every use names its set at compile time)
    auto bufsel = [&](auto I) -> Ops& {
      if constexpr (decltype(I)::value == 0) {
        return buf0;
      } else if constexpr (decltype(I)::value == 1) {
        return buf1;
      } else {
        return buf2;
      }
    };
#define LAT_BUF(i) bufsel(std::integral_constant<int, (i)>{})
// last statement of a switch arm that fills operand sets: arms that differ only in WHICH set they fill would otherwise be merged into one piece of code
// with the set's address chosen by a phi -- and a set whose address is a run-time value lives in scratch memory, every element of it
#define LAT_ARM(tag) asm volatile("; operand sets, arm " tag)
    auto issue = [&](uint32_t b, Ops& q) {
      const uint32_t at = (b * LPC + pl) * 8;
      load8s(X + at, q.x);
      load8s(Y + at, q.y);
      if (has_app) {
        load8s(A1 + at, q.a);
      }
    };
    // systematic (+ a-priori) and parity operands of the 8 steps of a block; ap: the a-priori values alone (extrinsic subtraction)
    auto prep = [&](const Ops& q, s2(&xs)[8], s2(&ys)[8], s2(&ap)[8]) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        xs[j] = q.x[j];
        ys[j] = q.y[j];
        ap[j] = splat(0);
        if (has_app) {
          ap[j] = q.a[j];
          xs[j] = AR::add(ap[j], xs[j]);
        }
      }
    };
    s2 o;
    // one backward step / one forward step without output at the residue (R0 + j) mod 3
    auto bstep = [&](auto RES, s2 x, s2 y) {
      constexpr int R = decltype(RES)::value;
      s2            go, gc;
      gammas<AR, R>(lk, x, y, go, gc);
      o = beta_step<AR, R>(o, go, gc);
    };
    auto astep = [&](auto RES, s2 x, s2 y) {
      constexpr int R = decltype(RES)::value;
      s2            go, gc;
      gammas<AR, R>(lk, x, y, go, gc);
      alpha_step<AR, R, false>(lk, o, o, go, gc);
    };
#define LAT_RES(R0, j) std::integral_constant<int, ((R0) + (j)) % 3> {}

    // ---- outputs of the 8 steps of block b from the branch sums of the forward recursion (t_o, t_c) and the backward metrics (bt): max-log-MAP
    // output, extrinsic subtraction, exchange of the 8 rows into the other decoder's a-priori array, decision source
    const uint32_t* lut  = dec1 ? p.deint : p.inter; // per (block, destination pair, step): row | source sub-blocks (turbo_host.cpp)
    uint32_t*       dst  = dec1 ? A2 : A1;
    const bool      last = (n + 1 == p.n_end) || crc_poly;
    auto emit = [&](auto R0C, auto FULLC, uint32_t b, int len, const s2(&bt)[8], const s2(&t_o)[8], const s2(&t_c)[8], const s2(&xs)[8], const s2(&ap)[8],
                    const uint32_t(&wrapj)[8], uint32_t trl_in) {
      constexpr int  R0   = decltype(R0C)::value;
      constexpr bool FULL = decltype(FULLC)::value;
      // max-log-MAP outputs of the 8 steps, STAGE BY STAGE across the steps: the eight reductions are independent, so every cross-lane move
      // reads a register written eight instructions earlier and needs no wait states (step by step, a fifth of the block was s_nop).
      // Even slots collect the data-bit-0 maximum, odd slots the data-bit-1 one: a lane keeps the candidate of its class, sends the other to
      // slot ^ 1, then ONE value per lane is reduced over slot ^ 2 and slot ^ 4.
      s2 w[8], u[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool q = lk.q[(R0 + j) % 3];
        w[j]         = AR::add_raw(bt[j], q ? t_o[j] : t_c[j]);
        u[j]         = AR::add_raw(bt[j], q ? t_c[j] : t_o[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        u[j] = from_u(partner<0>(to_u(u[j])));
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        w[j] = vmax(w[j], u[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        u[j] = from_u(partner<1>(to_u(w[j])));
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        w[j] = vmax(w[j], u[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        u[j] = from_u(partner<2>(to_u(w[j])));
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        w[j] = vmax(w[j], u[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        u[j] = from_u(partner<0>(to_u(w[j]))); // the other class' maximum
      }
      s2 kept = splat(0), keptraw = splat(0);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (FULL || j < len) {
          const s2       m1  = lk.odd ? w[j] : u[j];
          const s2       m0  = lk.odd ? u[j] : w[j];
          const s2       llr = AR::llr(AR::clean(m1), AR::clean(m0));
          const uint32_t k   = b * 8 + j;
          // decoder 1: ext1 - app1 (the a-priori it just used; zero in the first half iteration); decoder 2: ext2 - its systematic input
          s2 proc;
          if constexpr (dec1) {
            proc = AR::ex_sub(llr, ap[j], k == wrap_row);
          } else {
            proc = AR::ex_sub(llr, xs[j], AR::kIs8 && (wrapj[j] & 0xffffu) == wrap_row);
          }
          kept    = slot == j ? proc : kept; // lane (pair, j) keeps the output of step j
          keptraw = slot == j ? llr : keptraw;
        }
      }
      // exchange of the block's 8 rows: lane (pair p', step j) assembles the two values of its destination sub-blocks
      const uint32_t trl = trl_in;
      const uint32_t row = trl & 0xffffu, jlo = (trl >> 16) & 31u, jhi = (trl >> 21) & 31u;
      const int      a_l = gbase + (int)(jlo >> 1) * 8 + slot, c_l = gbase + (int)(jhi >> 1) * 8 + slot;
      auto           pick = [&](uint32_t v) {
        uint32_t a, c;
        read2(v, a_l, c_l, a, c);
        const uint32_t lo = (jlo & 1u) ? (a >> 16) : (a & 0xffffu);
        const uint32_t hi = (jhi & 1u) ? (c >> 16) : (c & 0xffffu);
        return lo | (hi << 16);
      };
      const uint32_t v   = pick(to_u(kept));
      const uint32_t dat = ((row >> 3) * LPC + pl) * 8 + (row & 7u);
      if (slot < len) {
        dst[dat] = v;
      }
      if (last) {
        // what tdec_decision_byte reads (turbodecoder.c:370-378), natural order: ext1 after decoder 1, the de-interleaved ext2 after decoder 2
        // (every lane takes part in the exchange: across two waves it has a barrier inside)
        uint32_t r, at;
        if constexpr (dec1) {
          r  = to_u(keptraw);
          at = (b * LPC + pl) * 8 + slot;
        } else {
          r  = pick(to_u(keptraw));
          at = dat;
        }
        if (slot < len) {
          D[at] = r;
        }
      }
    };

    // backward recursion, start: 40 steps on the head of every sub-block, then every estimate handed to the sub-block in front (turbodecoder_win.h:551-625)
    auto beta_start = [&]() {
      o = splat(-AR::kInf);
      // pass 0: 40 steps on the head of every sub-block, all states unknown.  Blocks 4 ... 0: sets 1 0 2 1 0
      {
        constexpr int WB = TD_WIN_OVERLAP / 8; // 5
        issue(WB - 1, LAT_BUF((WB - 1) % 3));
        issue(WB - 2, LAT_BUF((WB - 2) % 3));
        auto blk = [&](auto BC) {
          constexpr int b = decltype(BC)::value, set = b % 3, R0 = (2 * b) % 3;
          if constexpr (b >= 2) {
            issue(b - 2, LAT_BUF((b - 2) % 3));
          }
          s2 xs[8], ys[8], ap[8];
          prep(LAT_BUF(set), xs, ys, ap);
  #pragma unroll
          for (int j = 7; j >= 0; j--) {
            if ((R0 + j) % 3 == 0) {
              bstep(LAT_RES(0, 0), xs[j], ys[j]);
            } else if ((R0 + j) % 3 == 1) {
              bstep(LAT_RES(1, 0), xs[j], ys[j]);
            } else {
              bstep(LAT_RES(2, 0), xs[j], ys[j]);
            }
            if (AR::norm_at((uint32_t)b * 8 + j)) {
              o = normalise<AR>(o);
            }
          }
        };
        static_assert(WB == 5, "warm-up of 40 steps");
        blk(std::integral_constant<int, 4>{});
        blk(std::integral_constant<int, 3>{});
        blk(std::integral_constant<int, 2>{});
        blk(std::integral_constant<int, 1>{});
        blk(std::integral_constant<int, 0>{});
      }
      // hand every estimate (beta at step 0, labelling of residue 0) to the previous sub-block as its beta at step W (labelling of rW);
      // the last sub-block starts from the tail trellis
      {
        short tr[8];
        tail_trellis<AR>(xt, yt, tr);
        const int st = state_of(rW, slot); // state this slot must hold; at residue 0 slot == state
        short     tv = tr[0];
  #pragma unroll
        for (int i = 1; i < 8; i++) {
          tv = st == i ? tr[i] : tv;
        }
        uint32_t own, next;
        read2(to_u(o), gbase + pl * 8 + st, gbase + ((pl + 1) % LPC) * 8 + st, own, next);
        const uint32_t lo   = own >> 16;
        const uint32_t hi   = (pl == LPC - 1) ? (uint32_t)(uint16_t)tv : (next & 0xffffu);
        o                   = from_u(lo | (hi << 16));
        CK[nblk * G + li]   = to_u(o);
      }
    };
    // forward recursion, start: the last 40 steps of every sub-block, then every estimate handed to the sub-block behind (turbodecoder_win.h:684-750)
    auto alpha_start = [&]() {
      o = splat(-AR::kInf);
      {
        // warm-up: the last 40 steps of every sub-block, blocks w0 / 8 ... (W - 1) / 8, block b in set b mod 3
        const uint32_t w0 = long_sb - TD_WIN_OVERLAP;
        const uint32_t bl = (long_sb - 1) >> 3, b0 = w0 >> 3;
        auto           blk = [&](auto SET, uint32_t b) {
          constexpr int set = decltype(SET)::value, R0 = (2 * set) % 3;
          if (b + 2 <= bl) {
            issue(b + 2, LAT_BUF((set + 2) % 3));
          }
          s2 xs[8], ys[8], ap[8];
          prep(LAT_BUF(set), xs, ys, ap);
  #pragma unroll
          for (int j = 0; j < 8; j++) {
            const uint32_t k = b * 8 + j;
            if (k >= w0 && k < long_sb) {
              if ((R0 + j) % 3 == 0) {
                astep(LAT_RES(0, 0), xs[j], ys[j]);
              } else if ((R0 + j) % 3 == 1) {
                astep(LAT_RES(1, 0), xs[j], ys[j]);
              } else {
                astep(LAT_RES(2, 0), xs[j], ys[j]);
              }
              if (AR::norm_at(k - w0)) {
                o = normalise<AR>(o);
              }
            }
          }
        };
        switch (b0 % 3) {
          case 0: issue(b0, LAT_BUF(0)); issue(b0 + 1, LAT_BUF(1)); LAT_ARM("case0"); break;
          case 1: issue(b0, LAT_BUF(1)); issue(b0 + 1, LAT_BUF(2)); LAT_ARM("case1"); break;
          default: issue(b0, LAT_BUF(2)); issue(b0 + 1, LAT_BUF(0)); LAT_ARM("default"); break;
        }
        uint32_t b = b0;
        if (b % 3 == 1) {
          blk(std::integral_constant<int, 1>{}, b++);
        }
        if (b % 3 == 2 && b <= bl) {
          blk(std::integral_constant<int, 2>{}, b++);
        }
        for (; b + 2 <= bl; b += 3) {
          blk(std::integral_constant<int, 0>{}, b);
          blk(std::integral_constant<int, 1>{}, b + 1);
          blk(std::integral_constant<int, 2>{}, b + 2);
        }
        if (b <= bl) {
          blk(std::integral_constant<int, 0>{}, b++);
        }
        if (b <= bl) {
          blk(std::integral_constant<int, 1>{}, b++);
        }
      }
      // hand every estimate (alpha at step W, labelling of rW) to the next sub-block as its alpha at step 0 (slot == state); the first
      // sub-block starts in state 0
      {
        const int      src  = slot_of(rW, slot); // slot that holds state `slot` in the labelling of rW
        uint32_t own, prev;
        read2(to_u(o), gbase + pl * 8 + src, gbase + ((pl + LPC - 1) % LPC) * 8 + src, own, prev);
        const uint32_t lo   = (pl == 0) ? (uint32_t)(uint16_t)(short)(slot ? -AR::kInf : 0) : (prev >> 16);
        const uint32_t hi   = own & 0xffffu;
        o                   = from_u(lo | (hi << 16));
      }
    };

    if constexpr (DUAL) {
      // ================= both recursions at once, one per wave.  Blocks [0, hb) are the forward wave's first half and the backward wave's second,
      // blocks [hb, nblk) the other way round.  In its first half a wave files, per step, the metric the OTHER one's output needs: the forward
      // state in front of step k (what the branch sums start from), the backward metric of index k + 1 as the reference stores it (before
      // re-basing).  One barrier at the half-way mark; in its second half a wave forms the outputs of its blocks from its own running metric and
      // the filed rows, 8 steps at a time through emit() like the one-wave form -- without check-points and without re-deriving anything.
      extern __shared__ uint32_t drows[];
      const uint32_t hb = ((nblk >> 1) / 3) * 3; // (a multiple of three: the block loops below run in threes, operand set = block index mod 3)
      uint32_t*      RA = drows;                  // forward rows of the blocks [0, hb): (b, j) at (8 b + j) x 64 + lane
      uint32_t*      RB = drows + (size_t)hb * 512; // backward rows of the blocks [hb, nblk): (b, j) at (8 (b - hb) + j) x 64 + lane
      using T = std::true_type;
      using F = std::false_type;
      if (wave == 1) {
        beta_start();
        s2   bprev = o; // the backward metric of index k + 1 as filed, k the step about to be taken
        auto blk = [&](auto SET, auto FULLC, auto SECONDC, int b) {
          constexpr int  set = decltype(SET)::value, R0 = (2 * set) % 3;
          constexpr bool FULL = decltype(FULLC)::value, SECOND = decltype(SECONDC)::value;
          if (b >= 2) {
            issue((uint32_t)b - 2, LAT_BUF((set + 1) % 3));
          }
          s2 xs[8], ys[8], ap[8];
          prep(LAT_BUF(set), xs, ys, ap);
          if constexpr (!SECOND) {
#pragma unroll
            for (int j = 7; j >= 0; j--) {
              const uint32_t k = (uint32_t)b * 8 + j;
              if (FULL || k < long_sb) {
                RB[(((uint32_t)b - hb) * 8 + j) * 64 + li] = to_u(bprev);
                if ((R0 + j) % 3 == 0) {
                  bstep(LAT_RES(0, 0), xs[j], ys[j]);
                } else if ((R0 + j) % 3 == 1) {
                  bstep(LAT_RES(1, 0), xs[j], ys[j]);
                } else {
                  bstep(LAT_RES(2, 0), xs[j], ys[j]);
                }
                bprev = o;
                if (AR::norm_at(k)) {
                  o = normalise<AR>(o);
                }
              }
            }
          } else {
            const uint32_t trl = lut[((uint32_t)b * LPC + pl) * 8 + slot];
            uint32_t       wrapj[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // (8-bit only) destination rows of the 8 steps: the wrap flag of the extrinsic subtraction
            if (AR::kIs8 && !dec1 && wrap_row != 0xffffffffu) {
              load8(lut + ((uint32_t)b * LPC + pl) * 8, wrapj);
            }
            uint32_t       ar[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              ar[j] = RA[((uint32_t)b * 8 + j) * 64 + li];
            }
            s2 go[8], gc[8], bt[8], t_o[8], t_c[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              if ((R0 + j) % 3 == 0) {
                gammas<AR, 0>(lk, xs[j], ys[j], go[j], gc[j]);
                t_c[j] = AR::add_raw(from_u(partner<0>(ar[j])), gc[j]);
              } else if ((R0 + j) % 3 == 1) {
                gammas<AR, 1>(lk, xs[j], ys[j], go[j], gc[j]);
                t_c[j] = AR::add_raw(from_u(partner<1>(ar[j])), gc[j]);
              } else {
                gammas<AR, 2>(lk, xs[j], ys[j], go[j], gc[j]);
                t_c[j] = AR::add_raw(from_u(partner<2>(ar[j])), gc[j]);
              }
              t_o[j] = AR::add_raw(from_u(ar[j]), go[j]);
            }
#pragma unroll
            for (int j = 7; j >= 0; j--) {
              bt[j] = bprev;
              if ((R0 + j) % 3 == 0) {
                o = beta_step<AR, 0>(o, go[j], gc[j]);
              } else if ((R0 + j) % 3 == 1) {
                o = beta_step<AR, 1>(o, go[j], gc[j]);
              } else {
                o = beta_step<AR, 2>(o, go[j], gc[j]);
              }
              bprev = o;
              if (AR::norm_at((uint32_t)b * 8 + j)) {
                o = normalise<AR>(o);
              }
            }
            emit(std::integral_constant<int, R0>{}, T{}, (uint32_t)b, 8, bt, t_o, t_c, xs, ap, wrapj, trl);
          }
        };
        int b = (int)nblk - 1;
        switch (b % 3) {
          case 0: issue((uint32_t)b, LAT_BUF(0)); issue((uint32_t)b - 1, LAT_BUF(2)); LAT_ARM("case0"); break;
          case 1: issue((uint32_t)b, LAT_BUF(1)); issue((uint32_t)b - 1, LAT_BUF(0)); LAT_ARM("case1"); break;
          default: issue((uint32_t)b, LAT_BUF(2)); issue((uint32_t)b - 1, LAT_BUF(1)); LAT_ARM("default"); break;
        }
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        switch (b % 3) { // (the last block of a sub-block may be ragged: guarded code for that one)
          case 0: blk(S0{}, F{}, F{}, b--); break;
          case 1: blk(S1{}, F{}, F{}, b--); break;
          default: blk(S2{}, F{}, F{}, b--); break;
        }
        if (b % 3 == 1) {
          blk(S1{}, T{}, F{}, b--);
        }
        if (b % 3 == 0) {
          blk(S0{}, T{}, F{}, b--);
        }
        for (; b >= (int)hb + 2; b -= 3) { // (nblk >= 7 and hb <= nblk / 2: the blocks above never reach below hb; hb = 0 mod 3: this loop ends on it)
          blk(S2{}, T{}, F{}, b);
          blk(S1{}, T{}, F{}, b - 1);
          blk(S0{}, T{}, F{}, b - 2);
        }
        __syncthreads();
        for (; b >= 2; b -= 3) {
          blk(S2{}, T{}, T{}, b);
          blk(S1{}, T{}, T{}, b - 1);
          blk(S0{}, T{}, T{}, b - 2);
        }
      } else {
        alpha_start();
        auto blk = [&](auto SET, auto FULLC, auto SECONDC, uint32_t b) {
          constexpr int  set = decltype(SET)::value, R0 = (2 * set) % 3;
          constexpr bool FULL = decltype(FULLC)::value, SECOND = decltype(SECONDC)::value;
          const int      len = FULL ? 8 : ((long_sb - b * 8) < 8 ? (int)(long_sb - b * 8) : 8);
          if (b + 2 < nblk) {
            issue(b + 2, LAT_BUF((set + 2) % 3));
          }
          s2 xs[8], ys[8], ap[8];
          prep(LAT_BUF(set), xs, ys, ap);
          if constexpr (!SECOND) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
              RA[(b * 8 + j) * 64 + li] = to_u(o);
              if ((R0 + j) % 3 == 0) {
                astep(LAT_RES(0, 0), xs[j], ys[j]);
              } else if ((R0 + j) % 3 == 1) {
                astep(LAT_RES(1, 0), xs[j], ys[j]);
              } else {
                astep(LAT_RES(2, 0), xs[j], ys[j]);
              }
              if (AR::norm_at(b * 8 + j)) {
                o = normalise<AR>(o);
              }
            }
          } else {
            const uint32_t trl = lut[(b * LPC + pl) * 8 + slot];
            uint32_t       wrapj[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (AR::kIs8 && !dec1 && wrap_row != 0xffffffffu) {
              load8(lut + (b * LPC + pl) * 8, wrapj);
            }
            s2             go[8], gc[8], bt[8], t_o[8], t_c[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              bt[j] = from_u(RB[((b - hb) * 8 + j) * 64 + li]);
              if ((R0 + j) % 3 == 0) {
                gammas<AR, 0>(lk, xs[j], ys[j], go[j], gc[j]);
              } else if ((R0 + j) % 3 == 1) {
                gammas<AR, 1>(lk, xs[j], ys[j], go[j], gc[j]);
              } else {
                gammas<AR, 2>(lk, xs[j], ys[j], go[j], gc[j]);
              }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
              t_o[j] = t_c[j] = splat(0);
              if (FULL || j < len) {
                if ((R0 + j) % 3 == 0) {
                  alpha_branches<AR, 0>(o, go[j], gc[j], t_o[j], t_c[j]);
                } else if ((R0 + j) % 3 == 1) {
                  alpha_branches<AR, 1>(o, go[j], gc[j], t_o[j], t_c[j]);
                } else {
                  alpha_branches<AR, 2>(o, go[j], gc[j], t_o[j], t_c[j]);
                }
                if (AR::norm_at(b * 8 + j)) {
                  o = normalise<AR>(o);
                }
              }
            }
            emit(std::integral_constant<int, R0>{}, FULLC, b, len, bt, t_o, t_c, xs, ap, wrapj, trl);
          }
        };
        issue(0, LAT_BUF(0));
        issue(1, LAT_BUF(1));
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>;
        uint32_t b = 0;
        for (; b < hb; b += 3) {
          blk(S0{}, T{}, F{}, b);
          blk(S1{}, T{}, F{}, b + 1);
          blk(S2{}, T{}, F{}, b + 2);
        }
        __syncthreads();
        for (; b + 3 < nblk; b += 3) { // full blocks only: the last block of the sub-block is left to the guarded code below
          blk(S0{}, T{}, T{}, b);
          blk(S1{}, T{}, T{}, b + 1);
          blk(S2{}, T{}, T{}, b + 2);
        }
        if (b + 1 < nblk) {
          blk(S0{}, T{}, T{}, b++);
          if (b + 1 < nblk) {
            blk(S1{}, T{}, T{}, b++);
          }
        }
        switch (b % 3) { // b == nblk - 1
          case 0: blk(S0{}, F{}, T{}, b); break;
          case 1: blk(S1{}, F{}, T{}, b); break;
          default: blk(S2{}, F{}, T{}, b); break;
        }
      }
      __syncthreads();
      return;
    }
    // ================= backward recursion (turbodecoder_win.h:551-681)
    beta_start();
    // pass 1: whole sub-block, a check-point at every block boundary; blocks nblk - 1 ... 0, block b in set b mod 3, block b - 2 requested
    {
      // FULL: all 8 steps of the block exist (every block but a ragged last one): no per-step conditions, one scheduling region
      auto blk = [&](auto SET, auto FULLC, int b) {
        constexpr int  set = decltype(SET)::value, R0 = (2 * set) % 3;
        constexpr bool FULL = decltype(FULLC)::value;
        if (b >= 2) {
          issue((uint32_t)b - 2, LAT_BUF((set + 1) % 3));
        }
        s2 xs[8], ys[8], ap[8];
        prep(LAT_BUF(set), xs, ys, ap);
#pragma unroll
        for (int j = 7; j >= 0; j--) {
          const uint32_t k = (uint32_t)b * 8 + j;
          if (FULL || k < long_sb) {
            if ((R0 + j) % 3 == 0) {
              bstep(LAT_RES(0, 0), xs[j], ys[j]);
            } else if ((R0 + j) % 3 == 1) {
              bstep(LAT_RES(1, 0), xs[j], ys[j]);
            } else {
              bstep(LAT_RES(2, 0), xs[j], ys[j]);
            }
            if (j == 0 && b > 0) {
              CK[(uint32_t)b * G + li] = to_u(o);
            }
            if (AR::norm_at(k)) {
              o = normalise<AR>(o);
            }
          }
        }
      };
      int b = (int)nblk - 1;
      // (the sets of the two first blocks are only known at run time: their loads go through a switch once)
      switch (b % 3) {
        case 0: issue((uint32_t)b, LAT_BUF(0)); issue((uint32_t)b - 1, LAT_BUF(2)); LAT_ARM("case0"); break;
        case 1: issue((uint32_t)b, LAT_BUF(1)); issue((uint32_t)b - 1, LAT_BUF(0)); LAT_ARM("case1"); break;
        default: issue((uint32_t)b, LAT_BUF(2)); issue((uint32_t)b - 1, LAT_BUF(1)); LAT_ARM("default"); break;
      }
      // the first block may be ragged (W not a multiple of 8): guarded code for that one
      using T = std::true_type;
      using F = std::false_type;
      switch (b % 3) {
        case 0: blk(std::integral_constant<int, 0>{}, F{}, b--); break;
        case 1: blk(std::integral_constant<int, 1>{}, F{}, b--); break;
        default: blk(std::integral_constant<int, 2>{}, F{}, b--); break;
      }
      if (b % 3 == 1) {
        blk(std::integral_constant<int, 1>{}, T{}, b--);
      }
      if (b % 3 == 0) {
        blk(std::integral_constant<int, 0>{}, T{}, b--);
      }
      for (; b >= 2; b -= 3) {
        blk(std::integral_constant<int, 2>{}, T{}, b);
        blk(std::integral_constant<int, 1>{}, T{}, b - 1);
        blk(std::integral_constant<int, 0>{}, T{}, b - 2);
      }
    }
    __syncthreads(); // check-point stores before their loads

    // ================= forward recursion + LLR (turbodecoder_win.h:684-832)
    alpha_start();

    {
      uint32_t ckb[3], trb[3]; // check-point and exchange entry of a block, same three-set scheme
      auto     issue_aux = [&](uint32_t b, int set) {
        ckb[set] = CK[(b + 1) * G + li];
        trb[set] = lut[(b * LPC + pl) * 8 + slot];
      };
      auto blk = [&](auto SET, auto FULLC, uint32_t b) {
        constexpr int  set = decltype(SET)::value, R0 = (2 * set) % 3;
        constexpr bool FULL = decltype(FULLC)::value;
        const int      len = FULL ? 8 : ((long_sb - b * 8) < 8 ? (int)(long_sb - b * 8) : 8);
        if (b + 2 < nblk) {
          issue(b + 2, LAT_BUF((set + 2) % 3));
          issue_aux(b + 2, (set + 2) % 3);
        }
        s2 xs[8], ys[8], ap[8];
        prep(LAT_BUF(set), xs, ys, ap);
        uint32_t wrapj[8]; // (8-bit only) destination rows of the 8 steps: the wrap flag of the extrinsic subtraction
        if (AR::kIs8 && !dec1 && wrap_row != 0xffffffffu) {
          load8(lut + (b * LPC + pl) * 8, wrapj);
        }
        // branch metrics of the 8 steps: shared by the re-derived beta, alpha and the LLR
        s2 go[8], gc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if ((R0 + j) % 3 == 0) {
            gammas<AR, 0>(lk, xs[j], ys[j], go[j], gc[j]);
          } else if ((R0 + j) % 3 == 1) {
            gammas<AR, 1>(lk, xs[j], ys[j], go[j], gc[j]);
          } else {
            gammas<AR, 2>(lk, xs[j], ys[j], go[j], gc[j]);
          }
        }
        // re-derive beta(8 b + 1 ... 8 b + len) (the stored, pre-normalisation values) from the check-point
        s2 bt[8];
        s2 st = from_u(ckb[set]);
#pragma unroll
        for (int j = 7; j >= 0; j--) {
          bt[j] = st; // entry len - 1 is the check-point itself; the ones below it are overwritten
        }
#pragma unroll
        for (int j = 6; j >= 0; j--) {
          if (FULL || j <= len - 2) {
            const uint32_t idx = b * 8 + j + 2; // step index of the stored value we start from
            if ((FULL || idx != long_sb) && AR::norm_at(idx)) {
              st = normalise<AR>(st);
            }
            if ((R0 + j + 1) % 3 == 0) {
              st = beta_step<AR, 0>(st, go[j + 1], gc[j + 1]);
            } else if ((R0 + j + 1) % 3 == 1) {
              st = beta_step<AR, 1>(st, go[j + 1], gc[j + 1]);
            } else {
              st = beta_step<AR, 2>(st, go[j + 1], gc[j + 1]);
            }
            bt[j] = st;
          }
        }
        // forward recursion of the 8 steps: the two branch sums of every step are kept for the LLRs
        s2 t_o[8], t_c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          t_o[j] = t_c[j] = splat(0); // (steps behind a ragged sub-block's end: defined values, never kept)
          if (FULL || j < len) {
            if ((R0 + j) % 3 == 0) {
              alpha_branches<AR, 0>(o, go[j], gc[j], t_o[j], t_c[j]);
            } else if ((R0 + j) % 3 == 1) {
              alpha_branches<AR, 1>(o, go[j], gc[j], t_o[j], t_c[j]);
            } else {
              alpha_branches<AR, 2>(o, go[j], gc[j], t_o[j], t_c[j]);
            }
            if (AR::norm_at(b * 8 + j)) {
              o = normalise<AR>(o);
            }
          }
        }
        emit(std::integral_constant<int, R0>{}, FULLC, b, len, bt, t_o, t_c, xs, ap, wrapj, trb[set]);
      };
      issue(0, LAT_BUF(0));
      issue_aux(0, 0);
      issue(1, LAT_BUF(1));
      issue_aux(1, 1);
      using T = std::true_type;
      using F = std::false_type;
      uint32_t b = 0;
      for (; b + 3 < nblk; b += 3) { // full blocks only: the last block of the sub-block is left to the guarded code below
        blk(std::integral_constant<int, 0>{}, T{}, b);
        blk(std::integral_constant<int, 1>{}, T{}, b + 1);
        blk(std::integral_constant<int, 2>{}, T{}, b + 2);
      }
      if (b + 1 < nblk) {
        blk(std::integral_constant<int, 0>{}, T{}, b++);
        if (b + 1 < nblk) {
          blk(std::integral_constant<int, 1>{}, T{}, b++);
        }
      }
      switch (b % 3) { // b == nblk - 1
        case 0: blk(std::integral_constant<int, 0>{}, F{}, b); break;
        case 1: blk(std::integral_constant<int, 1>{}, F{}, b); break;
        default: blk(std::integral_constant<int, 2>{}, F{}, b); break;
      }
    }
#undef LAT_RES
#undef LAT_BUF
#undef LAT_ARM
    __syncthreads();
  };

  for (uint32_t n = p.n_begin; n < p.n_end; n++) {
    if (!(n & 1)) {
      half_iteration(std::true_type{}, n);
    } else {
      half_iteration(std::false_type{}, n);
    }
    if (crc_poly) {
      // decode_tb_cb (sch.c:420-454): hard bits + CRC after every half iteration; a block stops at its first match
      const bool     fin = n + 1 == p.n_end;
      const uint32_t crc = decide(!done, fin);
      if (!done) {
        noi++;
        done = crc == 0;
      }
      if ((G > 64 ? (done || !live) : (bool)__all(done || !live)) || fin) { // (a 128-lane block is alone in its workgroup: `done` is uniform)
        break;
      }
    }
  }
  if (!crc_poly) {
    decide(true, true);
  } else if (p.noi && live && li == 0) {
    p.noi[cb_raw]    = (int)noi;
    p.crc_ok[cb_raw] = done ? 1 : 0;
  }
}

} // namespace lat

uint32_t lat_ws_dwords(uint32_t K, int nb)
{
  return (lat::ws_dwords(K, nb) + 3u) & ~3u;
}

// two waves per code block, one per recursion (16 sub-blocks, 16-bit): LDS for the filed metric rows, 2 KB per 8-step block
size_t lat2_lds_bytes(uint32_t K)
{
  const uint32_t long_sb = K / 16, nblk = (long_sb + 7) / 8;
  return (size_t)nblk * 512 * sizeof(uint32_t);
}
template <class AR, bool ES>
static hipError_t launch_lat2_as(const WinParams& p, size_t lds, hipStream_t stream)
{
  static bool attr_set[kMaxDevices] = {};
  const int   dev = current_device(), di = dev >= 0 && dev < kMaxDevices ? dev : 0;
  if (!attr_set[di]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lat::tdec_lat_kernel<8, AR, ES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e != hipSuccess) {
      return e;
    }
    attr_set[di] = true;
  }
  hipLaunchKernelGGL((lat::tdec_lat_kernel<8, AR, ES, true>), dim3((unsigned)p.n_cb), dim3(128), lds, stream, p);
  return hipGetLastError();
}
hipError_t launch_lat2(bool arith8, const WinParams& p, hipStream_t stream)
{
  const bool   es  = p.crc_poly || p.desc;
  const size_t lds = lat2_lds_bytes(p.K);
  if (lds > 120 * 1024 || p.K < 16 * (TD_WIN_OVERLAP + 8)) {
    return hipErrorInvalidValue;
  }
  if (arith8) {
    return es ? launch_lat2_as<Ar8, true>(p, lds, stream) : launch_lat2_as<Ar8, false>(p, lds, stream);
  }
  return es ? launch_lat2_as<Ar16, true>(p, lds, stream) : launch_lat2_as<Ar16, false>(p, lds, stream);
}

// latency kernel: 16 / 8 sub-blocks with 16-bit arithmetic, 32 / 16 sub-blocks with 8-bit arithmetic (32 sub-blocks: two waves per block)
hipError_t launch_lat(int nb, bool arith8, const WinParams& p, hipStream_t stream)
{
  const bool es  = p.crc_poly || p.desc;
  const int  bpw = nb >= 16 ? 1 : 2;
  dim3       grid((unsigned)((p.n_cb + bpw - 1) / bpw));
#define LAUNCH(LPC, AR)                                                                                               \
  do {                                                                                                                \
    if (es) {                                                                                                         \
      hipLaunchKernelGGL((lat::tdec_lat_kernel<LPC, AR, true>), grid, dim3(LPC > 8 ? 8 * LPC : 64), 0, stream, p);    \
    } else {                                                                                                          \
      hipLaunchKernelGGL((lat::tdec_lat_kernel<LPC, AR, false>), grid, dim3(LPC > 8 ? 8 * LPC : 64), 0, stream, p);   \
    }                                                                                                                 \
  } while (0)
  if (!arith8 && nb == 16) {
    LAUNCH(8, Ar16);
  } else if (!arith8 && nb == 8) {
    LAUNCH(4, Ar16);
  } else if (arith8 && nb == 16) {
    LAUNCH(8, Ar8);
  } else if (arith8 && nb == 32) {
    LAUNCH(16, Ar8);
  } else {
    return hipErrorInvalidValue;
  }
#undef LAUNCH
  return hipGetLastError();
}

} // namespace turbo
} // namespace phyhip
