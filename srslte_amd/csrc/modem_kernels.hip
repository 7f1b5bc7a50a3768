// modem_kernels.hip -- soft demodulation fused with Gold-sequence descrambling (gfx950).
//
// Reference behaviour: lib/src/phy/modem/demod_soft.c (srsran_demod_soft_demodulate{,_s,_b}) followed by
// lib/src/phy/common/sequence.c:440-607 (srsran_sequence_apply_{f,s,c}), as pusch.c:419-443 / pdsch.c:693-744 chain them.
// The reference writes the LLRs, reads them back and writes them again; here one pass reads 8 B per symbol and writes
// Qm soft bits: the kernel is a pure HBM stream.
//
// One workgroup (256 lanes) = one tile of one job: 1024 symbols (lane l takes symbols l, l+256, l+512, l+768: every load
// instruction of a wave is one contiguous 512 B) or, without demodulation, 8192 soft bits.  The tile's <= 8192 chips of the
// scrambling sequence are produced by its first <= 16 lanes, each jumping to a 512-chip boundary with the tables of
// modem_device.h and running the two shift registers 16 chips at a time, into LDS as packed words.
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
__device__ __forceinline__ uint32_t step16_x1(uint32_t s)
{
  return (s >> 16) | ((((s >> 3) ^ s) & 0xffffu) << 15);
}
__device__ __forceinline__ uint32_t step16_x2(uint32_t s)
{
  return (s >> 16) | ((((s >> 3) ^ (s >> 2) ^ (s >> 1) ^ s) & 0xffffu) << 15);
}

__device__ __forceinline__ void make_chips(const Params& p, uint32_t seed, uint32_t bit0, uint32_t nbits, uint32_t* cb)
{
  const uint32_t nch = (nbits + MODEM_SEQ_CHUNK - 1) / MODEM_SEQ_CHUNK;
  if (threadIdx.x < nch) {
    const uint32_t j  = bit0 / MODEM_SEQ_CHUNK + threadIdx.x;
    uint32_t       s1 = p.x1_tab[j];
    uint32_t       s2 = 0;
    const uint32_t* col = p.x2_cols + (size_t)j * 31;
#pragma unroll
    for (int i = 0; i < 31; i++) {
      s2 ^= ((seed >> i) & 1u) ? col[i] : 0u;
    }
#pragma unroll
    for (int w = 0; w < 16; w++) {
      const uint32_t lo = (s1 ^ s2) & 0xffffu;
      s1                = step16_x1(s1);
      s2                = step16_x2(s2);
      const uint32_t hi = (s1 ^ s2) & 0xffffu;
      s1                = step16_x1(s1);
      s2                = step16_x2(s2);
      cb[threadIdx.x * 16 + w] = lo | (hi << 16);
    }
  }
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

// ---- one tile: 1024 symbols -------------------------------------------------------------------------------------------------
template <typename T, int MOD>
__device__ __forceinline__ void demod_tile(const Params& p, const Job& job, uint32_t tile, const uint32_t* cb)
{
  constexpr int QM   = MOD == 0 ? 1 : 2 * MOD;
  const float2* sym  = (const float2*)p.in + job.in_off;
  T*            out  = (T*)p.out + job.out_off;
  const bool    al   = (((uintptr_t)out) & 15u) == 0;
  const uint32_t s0  = tile * 1024u;
  float2        x[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t s = s0 + r * 256u + threadIdx.x;
    x[r]             = s < job.n ? sym[s] : make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t ls = r * 256u + threadIdx.x;
    const uint32_t s  = s0 + ls;
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
    if (job.scramble) {
      const uint32_t c = chips_at(cb, ls * QM);
#pragma unroll
      for (int i = 0; i < QM; i++) {
        v[i] = flip<T>(v[i], (c >> i) & 1u);
      }
    }
    store_bits<T, QM>(out + (size_t)s * QM, v, al);
  }
}

// ---- one tile of srsran_sequence_apply_*: 8192 soft bits ----------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pass_tile(const Params& p, const Job& job, uint32_t tile, const uint32_t* cb)
{
  constexpr uint32_t V  = 16 / sizeof(T);
  const T*           in = (const T*)p.in + job.in_off;
  T*                 out = (T*)p.out + job.out_off;
  const bool         al  = ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0;
  const uint32_t     e0  = tile * MODEM_TILE_BITS;
#pragma unroll
  for (uint32_t r = 0; r < MODEM_TILE_BITS / (256 * V); r++) {
    const uint32_t le = (r * 256u + threadIdx.x) * V;
    const uint32_t e  = e0 + le;
    if (e >= job.n) {
      continue;
    }
    const uint32_t c = job.scramble ? chips_at(cb, le) : 0u;
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
  __shared__ uint32_t cb[16 * 16 + 1];
  __shared__ Job      sjob;
  // job of this workgroup: binary search over the first-tile prefix
  if (threadIdx.x == 0) {
    if (p.jobs == nullptr) {
      sjob = p.single;
    } else {
      uint32_t lo = 0, hi = p.n_jobs - 1;
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (p.jobs[mid].tile0 <= blockIdx.x) {
          lo = mid;
        } else {
          hi = mid - 1;
        }
      }
      sjob = p.jobs[lo];
    }
    cb[256] = 0;
  }
  __syncthreads();
  const Job      job  = sjob;
  const uint32_t tile = blockIdx.x - job.tile0;
  if (tile >= job.ntiles) {
    return;
  }
  const uint32_t qm        = job.mod == 0 ? 1u : 2u * job.mod;
  const uint32_t tile_bits = job.mod == MOD_PASS ? MODEM_TILE_BITS : 1024u * qm;
  const uint32_t all_bits  = job.mod == MOD_PASS ? job.n : job.n * qm;
  if (job.scramble) {
    const uint32_t bit0 = tile * tile_bits;
    make_chips(p, job.seed, bit0, min(tile_bits, all_bits - bit0), cb);
    __syncthreads();
  }
  switch (job.mod) {
    case 0:
      demod_tile<T, 0>(p, job, tile, cb);
      break;
    case 1:
      demod_tile<T, 1>(p, job, tile, cb);
      break;
    case 2:
      demod_tile<T, 2>(p, job, tile, cb);
      break;
    case 3:
      demod_tile<T, 3>(p, job, tile, cb);
      break;
    case 4:
      demod_tile<T, 4>(p, job, tile, cb);
      break;
    default:
      pass_tile<T>(p, job, tile, cb);
      break;
  }
}

} // namespace

uint32_t tiles_of(uint32_t mod, uint32_t n)
{
  return mod == MOD_PASS ? ceil_div(n, MODEM_TILE_BITS) : ceil_div(n, 1024u);
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
