// modem_kernels.hip -- soft demodulation fused with Gold-sequence descrambling (gfx950).
//
// Reference behaviour: lib/src/phy/modem/demod_soft.c (srsran_demod_soft_demodulate{,_s,_b}) followed by
// lib/src/phy/common/sequence.c:440-607 (srsran_sequence_apply_{f,s,c}), as pusch.c:419-443 / pdsch.c:693-744 chain them.
// The reference writes the LLRs, reads them back and writes them again; here one pass reads 8 B per symbol and writes
// Qm soft bits: the kernel is a pure HBM stream.
//
// One workgroup (256 lanes) = one tile of one job: 2048 symbols, 512 consecutive ones per wave (lane l takes l, l+64, ...:
// every load instruction of a wave is one contiguous 512 B, 8 of them in flight) or, without demodulation, 16384 soft bits.
// Each wave produces its own <= 4096 chips of the scrambling sequence with its first <= 32 lanes, each jumping to a 128-chip boundary
// with the tables of modem_device.h and running the x2 shift register 16 chips at a time, into LDS as packed words.
//
// Arithmetic: the x86 reference mixes a SIMD body (round-to-nearest conversion of symbol * -SCALE, saturating packs,
// integer thresholds) with scalar tails (truncation, float thresholds); which rule applies depends on the symbol index
// and the job's length only, and both are replayed (oracle/orc_modem.c restates them).
#include "hip_common.h"
#include "modem_device.h"

namespace phyhip {
namespace modem {

namespace {

__device__ __forceinline__ int cvt_rn(float v) // _mm_cvtps_epi32
{
  return (v >= -2147483648.0f && v < 2147483648.0f) ? __float2int_rn(v) : (int)0x80000000;
}
__device__ __forceinline__ int cvt_tr(float v) // _mm_cvttps_epi32 / cvttss2si
{
  return (v >= -2147483648.0f && v < 2147483648.0f) ? __float2int_rz(v) : (int)0x80000000;
}
__device__ __forceinline__ int cvt_tr_d(double v)
{
  return (v >= -2147483648.0 && v < 2147483648.0) ? __double2int_rz(v) : (int)0x80000000;
}
__device__ __forceinline__ int sat16(int v)
{
  return min(max(v, -32768), 32767);
}
__device__ __forceinline__ int sat8(int v)
{
  return min(max(v, -128), 127);
}

template <typename T>
struct Lim; // integer soft-bit types: wrap to the type, saturate like the packs instructions, scale constants
template <>
struct Lim<int16_t> {
  static __device__ __forceinline__ int wrap(int v) { return (int)(int16_t)v; }
  static __device__ __forceinline__ int sat(int v) { return sat16(v); }
  static constexpr int                  S_BPSK = 100, S16 = 400, S64 = 700, S256 = 1000;
  static constexpr int                  GROUP = 4; // symbols per SIMD iteration of the 16/64-QAM bodies
};
template <>
struct Lim<int8_t> {
  static __device__ __forceinline__ int wrap(int v) { return (int)(int8_t)v; }
  static __device__ __forceinline__ int sat(int v) { return sat8(sat16(v)); }
  static constexpr int                  S_BPSK = 20, S16 = 30, S64 = 40, S256 = 50;
  static constexpr int                  GROUP = 8;
};

// ---- one symbol -> QM soft bits (integer types) ------------------------------------------------------------------------
template <typename T, int MOD>
__device__ __forceinline__ void demod_int(float re, float im, uint32_t idx, uint32_t n, const Consts& k, int* v)
{
  using L            = Lim<T>;
  constexpr bool B   = sizeof(T) == 1;
  const float    x[2] = {re, im};
  if (MOD == 0) {
    v[0] = L::wrap(cvt_tr_d((double)((float)(-L::S_BPSK) * (re + im)) * 0.70710678118654752440));
  } else if (MOD == 1) {
    // vector_simd.c:436-472 / 524-589: 16 values per iteration saturate, the scalar remainder wraps
    const uint32_t len = 2 * n, body = len - len % 16;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int t = cvt_tr(x[c] * (B ? k.qpsk_b : k.qpsk_s));
      v[c]        = (2 * idx + c < body) ? L::sat(t) : L::wrap(t);
    }
  } else if (MOD == 2) {
    const bool body = idx < n - n % L::GROUP;
    const int  off  = B ? k.o16_b : k.o16_s;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (body) {
        const int t = L::sat(cvt_rn(x[c] * (float)(-L::S16)));
        v[c]        = t;
        v[2 + c]    = L::wrap(L::wrap(abs(t)) - off);
      } else {
        const int y = L::wrap(cvt_tr((float)L::S16 * x[c]));
        v[c]        = L::wrap(-y);
        v[2 + c]    = L::wrap(cvt_tr((float)abs(y) - (B ? k.t16_tail_b : k.t16_tail_s)));
      }
    }
  } else if (MOD == 3) {
    const bool body = idx < n - n % L::GROUP;
    const int  o1 = B ? k.o64a_b : k.o64a_s, o2 = B ? k.o64b_b : k.o64b_s;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      int t, s;
      if (body) {
        t = L::sat(cvt_rn(x[c] * (float)(-L::S64)));
        s = t;
      } else {
        t = L::wrap(cvt_tr((float)L::S64 * x[c]));
        s = L::wrap(-t);
      }
      const int a1 = L::wrap(L::wrap(abs(t)) - o1);
      v[c]         = s;
      v[2 + c]     = a1;
      v[4 + c]     = L::wrap(L::wrap(abs(a1)) - o2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      float f  = -x[c];
      v[c]     = L::wrap(cvt_tr((float)L::S256 * f));
      f        = __fsub_rn(fabsf(f), k.c8);
      v[2 + c] = L::wrap(cvt_tr((float)L::S256 * f));
      f        = __fsub_rn(fabsf(f), k.c4);
      v[4 + c] = L::wrap(cvt_tr((float)L::S256 * f));
      f        = __fsub_rn(fabsf(f), k.c2);
      v[6 + c] = L::wrap(cvt_tr((float)L::S256 * f));
    }
  }
}

template <int MOD>
__device__ __forceinline__ void demod_float(float re, float im, const Consts& k, float* v)
{
  const float x[2] = {re, im};
  if (MOD == 0) {
    v[0] = (float)((double)(-(re + im)) * 0.70710678118654752440);
  } else if (MOD == 1) {
    v[0] = re * k.qpsk_f;
    v[1] = im * k.qpsk_f;
  } else if (MOD == 2) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      v[c]     = -x[c];
      v[2 + c] = __fsub_rn(fabsf(x[c]), k.f16);
    }
  } else if (MOD == 3) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      v[c]     = -x[c];
      v[2 + c] = __fsub_rn(fabsf(x[c]), k.f64a);
      v[4 + c] = __fsub_rn(fabsf(v[2 + c]), k.f64b);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      float f  = -x[c];
      v[c]     = f;
      f        = __fsub_rn(fabsf(f), k.c8);
      v[2 + c] = f;
      f        = __fsub_rn(fabsf(f), k.c4);
      v[4 + c] = f;
      f        = __fsub_rn(fabsf(f), k.c2);
      v[6 + c] = f;
    }
  }
}

// ---- scrambling chips of one tile -> LDS --------------------------------------------------------------------------------
// register = x(n)..x(n+30) in bits 0..30; 16 chips per step (the feedback taps reach back at most 3 chips)
__device__ __forceinline__ uint32_t step16_x2(uint32_t s)
{
  return (s >> 16) | ((((s >> 3) ^ (s >> 2) ^ (s >> 1) ^ s) & 0xffffu) << 15);
}

__device__ __forceinline__ void wave_sync_lds()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// chips bit0 .. bit0 + nbits - 1 (bit0 a multiple of 128, nbits <= MODEM_TILE_BITS / 4) of the sequence -> the wave's LDS strip, packed.
// One lane per 128 chips: x2 register at the chunk start = XOR of the table columns the seed selects, 8 steps of 16
// chips; the seed-independent x1 chips come packed from a table.  Waves work independently (no workgroup barrier): while
// one runs its shift registers the others stream.
__device__ __forceinline__ void make_chips(const uint32_t* x1_bits, const uint32_t* x2_cols, uint32_t seed, uint32_t bit0, uint32_t nbits, uint32_t* cbw)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t nch  = (nbits + MODEM_SEQ_CHUNK - 1) / MODEM_SEQ_CHUNK;
  if (lane < nch) {
    const uint32_t  j   = bit0 / MODEM_SEQ_CHUNK + lane;
    const uint32_t* col = x2_cols + (size_t)j * 31;
    const uint4     c1  = *(const uint4*)(x1_bits + (size_t)j * (MODEM_SEQ_CHUNK / 32));
    uint32_t        s2  = 0;
#pragma unroll
    for (int i = 0; i < 31; i++) {
      s2 ^= ((seed >> i) & 1u) ? col[i] : 0u;
    }
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t lo = s2 & 0xffffu;
      s2                = step16_x2(s2);
      const uint32_t hi = s2 & 0xffffu;
      s2                = step16_x2(s2);
      w[k]              = lo | (hi << 16);
    }
    *(uint4*)(cbw + lane * 4) = make_uint4(w[0] ^ c1.x, w[1] ^ c1.y, w[2] ^ c1.z, w[3] ^ c1.w);
  }
  wave_sync_lds();
}

__device__ __forceinline__ uint32_t chips_at(const uint32_t* cb, uint32_t off) // 32 chips starting at tile bit `off`
{
  const uint32_t w = off >> 5, sh = off & 31u;
  const uint64_t two = (uint64_t)cb[w] | ((uint64_t)cb[w + 1] << 32);
  return (uint32_t)(two >> sh);
}

template <typename T>
__device__ __forceinline__ T flip(T v, uint32_t bit)
{
  return bit ? (T)(-(int)v) : v;
}
template <>
__device__ __forceinline__ float flip<float>(float v, uint32_t bit)
{
  return __uint_as_float(__float_as_uint(v) ^ (bit << 31));
}

// ---- stores ---------------------------------------------------------------------------------------------------------------
template <typename T, int QM>
__device__ __forceinline__ void store_bits(T* dst, const T* v, bool aligned)
{
  constexpr int BYTES = QM * (int)sizeof(T);
  if (!aligned || BYTES < 4) {
#pragma unroll
    for (int i = 0; i < QM; i++) {
      dst[i] = v[i];
    }
    return;
  }
  uint32_t w[(BYTES + 3) / 4];
#pragma unroll
  for (int i = 0; i < (BYTES + 3) / 4; i++) {
    w[i] = 0;
  }
  if (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < QM; i++) {
      w[i] = __float_as_uint((float)v[i]);
    }
  } else if (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < QM; i++) {
      w[i / 2] |= ((uint32_t)(uint16_t)(int)v[i]) << (16 * (i & 1));
    }
  } else {
#pragma unroll
    for (int i = 0; i < QM; i++) {
      w[i / 4] |= ((uint32_t)(uint8_t)(int)v[i]) << (8 * (i & 3));
    }
  }
  if (BYTES == 4) {
    *(uint32_t*)dst = w[0];
  } else if (BYTES == 6) { // int8 64-QAM: 2-byte aligned
    uint16_t* d = (uint16_t*)dst;
    d[0]        = (uint16_t)w[0];
    d[1]        = (uint16_t)(w[0] >> 16);
    d[2]        = (uint16_t)w[1];
  } else if (BYTES == 8) {
    *(uint2*)dst = make_uint2(w[0], w[1]);
  } else if (BYTES == 12) {
    uint32_t* d = (uint32_t*)dst;
    d[0]        = w[0];
    d[1]        = w[1];
    d[2]        = w[2];
  } else if (BYTES == 16) {
    *(uint4*)dst = make_uint4(w[0], w[1], w[2], w[3]);
  } else if (BYTES == 24) {
    uint2* d = (uint2*)dst;
    d[0]     = make_uint2(w[0], w[1]);
    d[1]     = make_uint2(w[2], w[3]);
    d[2]     = make_uint2(w[4], w[5]);
  } else if (BYTES == 32) {
    uint4* d = (uint4*)dst;
    d[0]     = make_uint4(w[0], w[1], w[2], w[3]);
    d[1]     = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// Soft bits of 64 consecutive symbols whose size per symbol is not a power of two (6, 12, 24 bytes): the lanes' pieces go
// through a wave-private LDS strip (word stride 3 or 6: conflict-free) and leave as 16-byte stores, contiguous over the wave.
template <typename T, int QM>
__device__ __forceinline__ void store_bits_staged(T* wave_dst, const T* v, uint32_t* strip)
{
  constexpr int  BYTES = QM * (int)sizeof(T);
  const uint32_t lane  = threadIdx.x & 63u;
  if (BYTES == 6) {
    uint16_t* h = (uint16_t*)strip + lane * 3;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      h[i] = (uint16_t)((uint32_t)(uint8_t)(int)v[2 * i] | ((uint32_t)(uint8_t)(int)v[2 * i + 1] << 8));
    }
  } else {
    constexpr int W = BYTES / 4;
    uint32_t*     d = strip + lane * W;
#pragma unroll
    for (int i = 0; i < W; i++) {
      if (sizeof(T) == 4) {
        d[i] = __float_as_uint((float)v[i]);
      } else {
        d[i] = (uint32_t)(uint16_t)(int)v[2 * i] | ((uint32_t)(uint16_t)(int)v[2 * i + 1] << 16);
      }
    }
  }
  wave_sync_lds();
  constexpr int NQ = 64 * BYTES / 16;
#pragma unroll
  for (int k = 0; k < (NQ + 63) / 64; k++) {
    const uint32_t q = k * 64 + lane;
    if (q < NQ) {
      ((uint4*)wave_dst)[q] = ((const uint4*)strip)[q];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// ---- one tile: MODEM_TILE_SYMS symbols, a quarter per wave -------------------------------------------------------------------
template <typename T, int MOD>
__device__ __forceinline__ void demod_tile(const Params& p, const Job& job, uint32_t tile, uint32_t* cbw, uint32_t* strip)
{
  constexpr int  QM    = MOD == 0 ? 1 : 2 * MOD;
  constexpr int  BYTES = QM * (int)sizeof(T);
  constexpr bool STAGE = BYTES == 6 || BYTES == 12 || BYTES == 24;
  const float2*  sym   = (const float2*)p.in + job.in_off;
  T*             out   = (T*)p.out + job.out_off;
  const bool     al    = (((uintptr_t)out) & 15u) == 0;
  const uint32_t lane  = threadIdx.x & 63u;
  const uint32_t w0    = tile * MODEM_TILE_SYMS + (threadIdx.x >> 6) * (MODEM_TILE_SYMS / 4); // first symbol of this wave
  if (w0 >= job.n) {
    return;
  }
  constexpr int R = MODEM_TILE_SYMS / 256;
  float2        x[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t s = w0 + r * 64u + lane;
    x[r]             = s < job.n ? sym[s] : make_float2(0.f, 0.f);
  }
  if (job.scramble & 1u) { // the symbol loads are in flight while the first lanes run the shift registers
    make_chips(p.x1_bits, p.x2_cols, job.seed, w0 * QM, min((MODEM_TILE_SYMS / 4) * QM, (job.n - w0) * QM), cbw);
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t sw   = w0 + r * 64u;     // first symbol of this pass
    const uint32_t s    = sw + lane;
    const bool     full = sw + 64 <= job.n; // wave-uniform
    if (s >= job.n) {
      continue;
    }
    T v[QM];
    if constexpr (sizeof(T) == 4) {
      demod_float<MOD>(x[r].x, x[r].y, p.k, (float*)v);
    } else {
      int iv[QM];
      demod_int<T, MOD>(x[r].x, x[r].y, s, job.n, p.k, iv);
#pragma unroll
      for (int i = 0; i < QM; i++) {
        v[i] = (T)iv[i];
      }
    }
    if (job.scramble) { // bit 0: descramble, bit 1: negate everything (pdsch_nr.c:467)
      const uint32_t c = ((job.scramble & 1u) ? chips_at(cbw, (r * 64u + lane) * QM) : 0u) ^ ((job.scramble & 2u) ? ~0u : 0u);
#pragma unroll
      for (int i = 0; i < QM; i++) {
        v[i] = flip<T>(v[i], (c >> i) & 1u);
      }
    }
    if (job.il_rows) { // UL channel de-interleaver: symbol s of column-major order -> its row-major slot (modem_device.h)
      const uint32_t col = s / job.il_rows;
      store_bits<T, QM>(out + (size_t)((s - col * job.il_rows) * job.il_cols + col) * QM, v, false);
    } else if (STAGE && al && full) {
      store_bits_staged<T, QM>(out + (size_t)sw * QM, v, strip);
    } else {
      store_bits<T, QM>(out + (size_t)s * QM, v, al);
    }
  }
}

// ---- one tile of srsran_sequence_apply_*: MODEM_TILE_BITS soft bits, a quarter per wave ------------------------------------------
template <typename T>
__device__ __forceinline__ void pass_tile(const Params& p, const Job& job, uint32_t tile, uint32_t* cbw)
{
  constexpr uint32_t V    = 16 / sizeof(T);
  const T*           in   = (const T*)p.in + job.in_off;
  T*                 out  = (T*)p.out + job.out_off;
  const bool         al   = ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0;
  const uint32_t     lane = threadIdx.x & 63u;
  const uint32_t     w0   = tile * MODEM_TILE_BITS + (threadIdx.x >> 6) * (MODEM_TILE_BITS / 4);
  if (w0 >= job.n) {
    return;
  }
  if (job.scramble & 1u) {
    make_chips(p.x1_bits, p.x2_cols, job.seed, w0, min(MODEM_TILE_BITS / 4, job.n - w0), cbw);
  }
#pragma unroll
  for (uint32_t r = 0; r < MODEM_TILE_BITS / 4 / (64 * V); r++) {
    const uint32_t le = (r * 64u + lane) * V;
    const uint32_t e  = w0 + le;
    if (e >= job.n) {
      continue;
    }
    const uint32_t c = ((job.scramble & 1u) ? chips_at(cbw, le) : 0u) ^ ((job.scramble & 2u) ? ~0u : 0u);
    if (al && e + V <= job.n) {
      union {
        uint4 q;
        T     t[V];
      } u;
      u.q = *(const uint4*)(in + e);
#pragma unroll
      for (uint32_t i = 0; i < V; i++) {
        u.t[i] = flip<T>(u.t[i], (c >> i) & 1u);
      }
      *(uint4*)(out + e) = u.q;
    } else {
      for (uint32_t i = 0; i < V && e + i < job.n; i++) {
        out[e + i] = flip<T>(in[e + i], (c >> i) & 1u);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void modem_kernel(const Params p)
{
  __shared__ __attribute__((aligned(16))) uint32_t cb[4][MODEM_TILE_BITS / 128 + 4];
  __shared__ __attribute__((aligned(16))) uint32_t strips[4][384];
  __shared__ Job      sjob;
  // job of this workgroup: the host lists the job of every tile
  const uint32_t lo = p.jobs ? p.tile_job[blockIdx.x] : 0u;
  if (threadIdx.x < sizeof(Job) / 4) {
    const uint32_t* src = p.jobs ? (const uint32_t*)(p.jobs + lo) : (const uint32_t*)&p.single;
    ((uint32_t*)&sjob)[threadIdx.x] = src[threadIdx.x];
  }
  if ((threadIdx.x & 63u) == 0) {
    cb[threadIdx.x >> 6][MODEM_TILE_BITS / 128] = 0; // chips_at reads one word past the last one
  }
  __syncthreads();
  const Job      job  = sjob;
  const uint32_t tile = blockIdx.x - job.tile0;
  if (tile >= job.ntiles) {
    return;
  }
  uint32_t* cbw   = cb[threadIdx.x >> 6];
  uint32_t* strip = strips[threadIdx.x >> 6];
  switch (job.mod) {
    case 0:
      demod_tile<T, 0>(p, job, tile, cbw, strip);
      break;
    case 1:
      demod_tile<T, 1>(p, job, tile, cbw, strip);
      break;
    case 2:
      demod_tile<T, 2>(p, job, tile, cbw, strip);
      break;
    case 3:
      demod_tile<T, 3>(p, job, tile, cbw, strip);
      break;
    case 4:
      demod_tile<T, 4>(p, job, tile, cbw, strip);
      break;
    default:
      pass_tile<T>(p, job, tile, cbw);
      break;
  }
}

// ---- modulator: one workgroup = MODEM_TILE_SYMS symbols, a quarter per wave; the wave's chips as in demod_tile
template <int MOD>
__device__ __forceinline__ void mod_tile(const ModParams& p, uint32_t tile, uint32_t* cbw)
{
  constexpr int  QM   = MOD == 0 ? 1 : 2 * MOD;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w0   = tile * MODEM_TILE_SYMS + (threadIdx.x >> 6) * (MODEM_TILE_SYMS / 4);
  if (w0 >= p.n) {
    return;
  }
  if (p.scramble) {
    make_chips(p.x1_bits, p.x2_cols, p.seed, w0 * QM, min((MODEM_TILE_SYMS / 4) * QM, (p.n - w0) * QM), cbw);
  }
  const float2*  tab    = p.table + mod_table_offset(MOD);
  const uint32_t nbytes = (p.n * QM + 7) / 8;
#pragma unroll
  for (int r = 0; r < (int)(MODEM_TILE_SYMS / 256); r++) {
    const uint32_t s = w0 + r * 64u + lane;
    if (s >= p.n) {
      continue;
    }
    const uint32_t b  = s * QM, by = b >> 3;
    const uint32_t hi = p.bits[by], lo = by + 1 < nbytes ? p.bits[by + 1] : 0u;
    uint32_t       v  = (((hi << 8) | lo) >> (16 - QM - (b & 7u))) & ((1u << QM) - 1u); // bit 0 of the symbol = MSB of v
    if (p.scramble) {
      const uint32_t c = chips_at(cbw, (r * 64u + lane) * QM); // chip i of the symbol in bit i
#pragma unroll
      for (int i = 0; i < QM; i++) {
        v ^= ((c >> i) & 1u) << (QM - 1 - i);
      }
    }
    float2 o = tab[v];
    if (p.scale != 1.0f) {
      o.x = __fmul_rn(o.x, p.scale);
      o.y = __fmul_rn(o.y, p.scale);
    }
    p.out[s] = o;
  }
}

__global__ __launch_bounds__(256) void mod_kernel(const ModParams pin)
{
  __shared__ __attribute__((aligned(16))) uint32_t cb[4][MODEM_TILE_BITS / 128 + 4];
  if ((threadIdx.x & 63u) == 0) {
    cb[threadIdx.x >> 6][MODEM_TILE_BITS / 128] = 0;
  }
  __shared__ ModJob sjob;
  if (pin.jobs && threadIdx.x < sizeof(ModJob) / 4) {
    ((uint32_t*)&sjob)[threadIdx.x] = ((const uint32_t*)(pin.jobs + pin.tile_job[blockIdx.x]))[threadIdx.x];
  }
  __syncthreads();
  ModParams p    = pin;
  uint32_t  tile = blockIdx.x;
  if (pin.jobs) {
    const ModJob j = sjob;
    p.bits += j.bits_off;
    p.out += j.out_off;
    p.mod = j.mod, p.n = j.n, p.seed = j.seed, p.scramble = j.scramble, p.scale = j.scale;
    tile -= j.tile0;
  }
  uint32_t* cbw = cb[threadIdx.x >> 6];
  switch (p.mod) {
    case 0:
      mod_tile<0>(p, tile, cbw);
      break;
    case 1:
      mod_tile<1>(p, tile, cbw);
      break;
    case 2:
      mod_tile<2>(p, tile, cbw);
      break;
    case 3:
      mod_tile<3>(p, tile, cbw);
      break;
    default:
      mod_tile<4>(p, tile, cbw);
      break;
  }
}

// ---- scrambling of byte-packed bits: one workgroup = MODEM_TILE_BITS bits, a quarter per wave, two words per lane
__global__ __launch_bounds__(256) void scramble_packed_kernel(const uint32_t* in, uint32_t* out, uint32_t nwords, uint32_t seed, const uint32_t* x1_bits,
                                                              const uint32_t* x2_cols)
{
  __shared__ __attribute__((aligned(16))) uint32_t cb[4][MODEM_TILE_BITS / 128 + 4];
  if ((threadIdx.x & 63u) == 0) {
    cb[threadIdx.x >> 6][MODEM_TILE_BITS / 128] = 0;
  }
  __syncthreads();
  uint32_t*      cbw  = cb[threadIdx.x >> 6];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w0   = (blockIdx.x * MODEM_TILE_BITS + (threadIdx.x >> 6) * (MODEM_TILE_BITS / 4)) / 32u; // first word of this wave
  if (w0 >= nwords) {
    return;
  }
  make_chips(x1_bits, x2_cols, seed, w0 * 32u, min(MODEM_TILE_BITS / 4, (nwords - w0) * 32u), cbw);
#pragma unroll
  for (uint32_t r = 0; r < MODEM_TILE_BITS / 4 / 32 / 64; r++) {
    const uint32_t lw = r * 64u + lane;
    if (w0 + lw < nwords) {
      // chip i of the word sits in bit i; the packed bits are MSB first inside every byte: reverse the word, then its bytes
      const uint32_t c = __builtin_bswap32(__brev(cbw[lw]));
      out[w0 + lw]     = in[w0 + lw] ^ c;
    }
  }
}

// ---- single-antenna ZF / MMSE equaliser: srsran_predecoding_single (mimo/precoding.c:196-392), the float formulas of its AVX
// body: x = (y conj(h)) / (|h|^2 + noise) * (1 / scaling); two symbols (one dwordx4 of y and of h) per lane
__global__ __launch_bounds__(256) void eq_kernel(const float4* y, const float4* h, float4* x, float2* csi, uint32_t n, float inv_scaling,
                                                 float noise, int add_noise)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x; // pair index
  if (2 * i >= n) {
    return;
  }
  float4 yy, hh;
  if (2 * i + 1 < n) {
    yy = y[i];
    hh = h[i];
  } else { // odd tail
    const float2 a = ((const float2*)y)[2 * i], b = ((const float2*)h)[2 * i];
    yy             = make_float4(a.x, a.y, 0.f, 0.f);
    hh             = make_float4(b.x, b.y, 1.f, 0.f);
  }
  float c0 = __fadd_rn(__fmul_rn(hh.x, hh.x), __fmul_rn(hh.y, hh.y));
  float c1 = __fadd_rn(__fmul_rn(hh.z, hh.z), __fmul_rn(hh.w, hh.w));
  if (add_noise) {
    c0 = __fadd_rn(c0, noise);
    c1 = __fadd_rn(c1, noise);
  }
  float4 o;
  o.x = __fmul_rn(__fdiv_rn(__fadd_rn(__fmul_rn(yy.x, hh.x), __fmul_rn(yy.y, hh.y)), c0), inv_scaling);
  o.y = __fmul_rn(__fdiv_rn(__fsub_rn(__fmul_rn(yy.y, hh.x), __fmul_rn(yy.x, hh.y)), c0), inv_scaling);
  o.z = __fmul_rn(__fdiv_rn(__fadd_rn(__fmul_rn(yy.z, hh.z), __fmul_rn(yy.w, hh.w)), c1), inv_scaling);
  o.w = __fmul_rn(__fdiv_rn(__fsub_rn(__fmul_rn(yy.w, hh.z), __fmul_rn(yy.z, hh.w)), c1), inv_scaling);
  if (2 * i + 1 < n) {
    x[i] = o;
    if (csi) {
      csi[i] = make_float2(c0, c1);
    }
  } else {
    ((float2*)x)[2 * i] = make_float2(o.x, o.y);
    if (csi) {
      ((float*)csi)[2 * i] = c0;
    }
  }
}


// the same over several grants in one launch: contiguous runs of symbol PAIRS, each with its own noise estimate (the grants of a TTI, chan_host.cpp).
// jobs[j].end_pair = first pair behind grant j (ascending); every grant holds an even number of symbols.
__global__ __launch_bounds__(256) void eq_jobs_kernel(const float4* y, const float4* h, float4* x, const EqJob* jobs, uint32_t n_jobs, uint32_t n_pairs, float inv_scaling)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_pairs) {
    return;
  }
  uint32_t lo = 0, hi = n_jobs - 1; // smallest j with end_pair[j] > i
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobs[mid].end_pair > i) {
      hi = mid;
    } else {
      lo = mid + 1;
    }
  }
  const float  noise = jobs[lo].noise;
  const bool   add_noise = jobs[lo].add_noise != 0;
  const float4 yy = y[i], hh = h[i];
  float        c0 = __fadd_rn(__fmul_rn(hh.x, hh.x), __fmul_rn(hh.y, hh.y));
  float        c1 = __fadd_rn(__fmul_rn(hh.z, hh.z), __fmul_rn(hh.w, hh.w));
  if (add_noise) {
    c0 = __fadd_rn(c0, noise);
    c1 = __fadd_rn(c1, noise);
  }
  float4 o;
  o.x  = __fmul_rn(__fdiv_rn(__fadd_rn(__fmul_rn(yy.x, hh.x), __fmul_rn(yy.y, hh.y)), c0), inv_scaling);
  o.y  = __fmul_rn(__fdiv_rn(__fsub_rn(__fmul_rn(yy.y, hh.x), __fmul_rn(yy.x, hh.y)), c0), inv_scaling);
  o.z  = __fmul_rn(__fdiv_rn(__fadd_rn(__fmul_rn(yy.z, hh.z), __fmul_rn(yy.w, hh.w)), c1), inv_scaling);
  o.w  = __fmul_rn(__fdiv_rn(__fsub_rn(__fmul_rn(yy.w, hh.z), __fmul_rn(yy.z, hh.w)), c1), inv_scaling);
  x[i] = o;
}

} // namespace

hipError_t launch_eq(const void* y, const void* h, void* x, float* csi, uint32_t n, float scaling, float noise, hipStream_t stream)
{
  if (n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(eq_kernel, dim3(ceil_div(ceil_div(n, 2u), 256u)), dim3(256), 0, stream, (const float4*)y, (const float4*)h, (float4*)x,
                     (float2*)csi, n, 1.0f / scaling, noise, (csi != nullptr || noise > 0.f) ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_eq_jobs(const void* y, const void* h, void* x, const EqJob* jobs, uint32_t n_jobs, uint32_t n_pairs, float scaling, hipStream_t stream)
{
  if (n_jobs == 0 || n_pairs == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(eq_jobs_kernel, dim3(ceil_div(n_pairs, 256u)), dim3(256), 0, stream, (const float4*)y, (const float4*)h, (float4*)x, jobs, n_jobs, n_pairs,
                     1.0f / scaling);
  return hipGetLastError();
}

hipError_t launch_mod(const ModParams& p, hipStream_t stream)
{
  if (p.n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(mod_kernel, dim3(ceil_div(p.n, MODEM_TILE_SYMS)), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_mod_jobs(const ModParams& p, uint32_t n_tiles, hipStream_t stream)
{
  if (n_tiles == 0 || !p.jobs || !p.tile_job) {
    return n_tiles ? hipErrorInvalidValue : hipSuccess;
  }
  hipLaunchKernelGGL(mod_kernel, dim3(n_tiles), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_scramble_packed(const uint8_t* in, uint8_t* out, uint32_t nbits, uint32_t seed, const uint32_t* x1_bits, const uint32_t* x2_cols, hipStream_t stream)
{
  const uint32_t nwords = ceil_div(nbits, 32u);
  if (nwords == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(scramble_packed_kernel, dim3(ceil_div(nwords * 32u, MODEM_TILE_BITS)), dim3(256), 0, stream, (const uint32_t*)in, (uint32_t*)out, nwords, seed,
                     x1_bits, x2_cols);
  return hipGetLastError();
}

uint32_t tiles_of(uint32_t mod, uint32_t n)
{
  return mod == MOD_PASS ? ceil_div(n, MODEM_TILE_BITS) : ceil_div(n, MODEM_TILE_SYMS);
}

hipError_t launch(const Params& p, hipStream_t stream)
{
  if (p.n_tiles == 0) {
    return hipSuccess;
  }
  switch (p.llr_type) {
    case LLR_I16:
      hipLaunchKernelGGL(modem_kernel<int16_t>, dim3(p.n_tiles), dim3(256), 0, stream, p);
      break;
    case LLR_I8:
      hipLaunchKernelGGL(modem_kernel<int8_t>, dim3(p.n_tiles), dim3(256), 0, stream, p);
      break;
    default:
      hipLaunchKernelGGL(modem_kernel<float>, dim3(p.n_tiles), dim3(256), 0, stream, p);
      break;
  }
  return hipGetLastError();
}

} // namespace modem
} // namespace phyhip
