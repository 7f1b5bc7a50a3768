// tcod_kernels.hip -- LTE turbo encoder (TS 36.212 5.1.3.2; lib/src/phy/fec/turbo/turbocoder.c:76-185) and the transmit side of a
// transport block (sch.c:230-330 encode_tb: CRC24A, segmentation, CRC24B, turbo coding, rate matching, concatenation) for gfx950.
//
// One wave = one code block.  The two recursive convolutional encoders are linear over GF(2): every lane runs its stretch of
// the block from the zero state, the true entry states follow from a scan over the 64 lanes (state' = A^len state + v; A has
// period 7), and a second run from the true entry state emits the parity bits.  The CRCs use the same splitting: a bit-serial
// checksum per stretch, shifted into place by x^(bits behind it) mod g and XOR-ed over the wave.
#include "hip_common.h"
#include "tcod_device.h"

namespace phyhip {
namespace tcod {

namespace {

#define TX_NULL 100u // turbocoder.h:41 SRSRAN_TX_NULL

// registers r0 (newest), r1, r2 in bits 0..2; turbocoder.c:118-124
__device__ __forceinline__ uint32_t rsc_step(uint32_t s, uint32_t bit, uint32_t* out)
{
  const uint32_t r0 = s & 1u, r1 = (s >> 1) & 1u, r2 = (s >> 2) & 1u;
  const uint32_t in = bit ^ r2 ^ r1;
  *out              = r2 ^ r0 ^ in;
  return in | (r0 << 1) | (r1 << 2);
}

__device__ __forceinline__ uint32_t rsc_idle(uint32_t s, uint32_t n) // n zero-input steps (period 7)
{
  n %= 7u;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t o;
    s = rsc_step(s, 0u, &o);
  }
  return s;
}

struct Qpp { // PI(i) = (f1 i + f2 i^2) mod K, walked incrementally
  uint32_t pi, g, K, step2;
  __device__ __forceinline__ void start(uint32_t i, uint32_t K_, uint32_t f1, uint32_t f2)
  {
    K     = K_;
    pi    = (uint32_t)(((uint64_t)f1 * i + (uint64_t)f2 * i % K * i) % K);
    g     = (uint32_t)((f1 + (uint64_t)f2 * (2ull * i + 1ull)) % K); // PI(i+1) - PI(i)
    step2 = (2u * f2) % K;
  }
  __device__ __forceinline__ void next()
  {
    pi += g;
    pi -= pi >= K ? K : 0u;
    g += step2;
    g -= g >= K ? K : 0u;
  }
};

// d: the natural code word in LDS, 3 K + 12 bytes: d[3 i] = input bit i (0 / 1 / TX_NULL) on entry; d[3 i + 1], d[3 i + 2] receive
// the two parity bits, d[3 K ..] the 12 tail bits.  All 64 lanes of the wave call this.
__device__ __forceinline__ void encode_block(uint8_t* d, uint32_t K, uint32_t f1, uint32_t f2)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t L    = (K + 63u) / 64u;
  const uint32_t i0 = min(lane * L, K), i1 = min(i0 + L, K);
  uint32_t       fin[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    // run from the zero state
    uint32_t s = 0, o;
    Qpp      q;
    q.start(i0, K, f1, f2);
    for (uint32_t i = i0; i < i1; i++) {
      uint32_t b = d[3u * (e == 0 ? i : q.pi)];
      s          = rsc_step(s, b == TX_NULL ? 0u : b, &o);
      q.next();
    }
    // true entry states: lane l+1 enters where lane l leaves
    uint32_t s_in = 0, s_out = s; // lane 0 enters at zero: its zero-state run is already the truth
    for (uint32_t l = 0; l < 63u; l++) {
      const uint32_t v = __shfl(s_out, l);
      if (lane == l + 1) {
        s_in  = v;
        s_out = rsc_idle(v, i1 - i0) ^ s;
      }
    }
    fin[e] = __shfl(s_out, 63);
    // second run, from the true entry state
    s = s_in;
    q.start(i0, K, f1, f2);
    for (uint32_t i = i0; i < i1; i++) {
      uint32_t b       = d[3u * (e == 0 ? i : q.pi)];
      s                = rsc_step(s, b == TX_NULL ? 0u : b, &o);
      d[3u * i + 1 + e] = (uint8_t)o;
      q.next();
    }
  }
  if (lane < 2) { // turbocoder.c:150-185: three (systematic, parity) pairs per constituent encoder
    uint32_t s = fin[lane];
    for (int j = 0; j < 3; j++) {
      const uint32_t bit = ((s >> 2) ^ (s >> 1)) & 1u;
      uint32_t       o;
      s                        = rsc_step(s, bit, &o);
      d[3u * K + lane * 6 + 2 * j]     = (uint8_t)bit;
      d[3u * K + lane * 6 + 2 * j + 1] = (uint8_t)o;
    }
  }
}

__global__ __launch_bounds__(64) void encode_kernel(const EncParams p)
{
  extern __shared__ uint8_t sm[]; // the natural code word d
  const uint32_t K   = p.K;
  const uint8_t* in  = p.in + (size_t)blockIdx.x * p.in_stride;
  uint8_t*       out = p.out + (size_t)blockIdx.x * p.out_stride;
  for (uint32_t i = threadIdx.x; i < K; i += 64) {
    sm[3u * i] = in[i];
  }
  __syncthreads();
  encode_block(sm, K, p.f1, p.f2);
  __syncthreads();
  // turbocoder.c:109-147: systematic raw, first parity TX_NULL where the input is, second parity always a bit
  for (uint32_t t = threadIdx.x; t < 3 * K + 12; t += 64) {
    uint8_t v = sm[t];
    if (t < 3 * K && t % 3u == 1 && sm[t - 1] == TX_NULL) {
      v = (uint8_t)TX_NULL;
    }
    out[t] = v;
  }
}

// ---- CRC helpers (crc.c:92-140: MSB first, zero initial state, 24-bit generators) ----------------------------------------------
__device__ __forceinline__ uint32_t crc24_bit(uint32_t c, uint32_t bit, uint32_t poly)
{
  return ((c << 1) & 0xffffffu) ^ ((((c >> 23) ^ bit) & 1u) ? poly : 0u);
}
__device__ __forceinline__ uint32_t mulmod24(uint32_t a, uint32_t m, uint32_t poly) // a(x) m(x) mod g(x)
{
  uint32_t r = 0;
  for (int i = 23; i >= 0; i--) {
    r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? poly : 0u);
    r ^= ((m >> i) & 1u) ? a : 0u;
  }
  return r;
}
__device__ __forceinline__ uint32_t xpow24(uint32_t n, uint32_t poly) // x^n mod g(x)
{
  uint32_t r = 1, b = 2; // b = x
  while (n) {
    if (n & 1u) {
      r = mulmod24(r, b, poly);
    }
    b = mulmod24(b, b, poly);
    n >>= 1;
  }
  return r;
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v ^= __shfl_xor(v, off);
  }
  return v;
}

#define CRC24A_POLY 0x864CFBu // crc.h:41-42 without the x^24 term
#define CRC24B_POLY 0x800063u

// CRC24A of the payload bytes of one transport block: one wave per block
__global__ __launch_bounds__(64) void tb_crc24a_kernel(const TbParams p)
{
  const TbCrcJob  job = p.tbs[blockIdx.x];
  const uint8_t*  d   = p.data + job.src_byte;
  const uint32_t  L   = (job.n_bytes + 63u) / 64u;
  const uint32_t  i0 = min((threadIdx.x & 63u) * L, job.n_bytes), i1 = min(i0 + L, job.n_bytes);
  uint32_t        c = 0;
  for (uint32_t i = i0; i < i1; i++) {
    const uint32_t b = d[i];
#pragma unroll
    for (int k = 7; k >= 0; k--) {
      c = crc24_bit(c, (b >> k) & 1u, CRC24A_POLY);
    }
  }
  // (the multiplier x^(8 bytes behind this lane's stretch) comes from the host: a square-and-multiply per lane was two fifths of this kernel)
  c = mulmod24(c, p.crc_mult[job.crc_mult_row * 64u + (threadIdx.x & 63u)], CRC24A_POLY);
  c = wave_xor(c);
  if (threadIdx.x == 0) {
    p.tb_crc[blockIdx.x] = c;
  }
}

// one code block of a transport block: gather its bits, append the CRCs, encode, rate-match into the packed output
__global__ __launch_bounds__(64) void tb_encode_kernel(const TbParams p)
{
  extern __shared__ uint8_t sm[]; // the natural code word d, as in the encoder kernel
  const TbCbJob  job  = p.cbs[blockIdx.x];
  const uint32_t K    = job.K;
  const uint32_t lane = threadIdx.x;
  // payload bits of this block (sch.c:262-286: bytes of the transport block, MSB first)
  for (uint32_t i = lane; i < job.n_src_bits; i += 64) {
    const uint32_t b = job.src_bit + i;
    sm[3u * i]       = (p.data[b >> 3] >> (7u - (b & 7u))) & 1u;
  }
  uint32_t n = job.n_src_bits;
  if (job.tb_crc != 0xffffffffu) { // the transport-block CRC closes the last block (sch.c:268-273)
    const uint32_t crc = p.tb_crc[job.tb_crc];
    if (lane < 24) {
      sm[3u * (n + lane)] = (crc >> (23u - lane)) & 1u;
    }
    n += 24;
  }
  __syncthreads();
  if (job.crc24b) { // sch.c:276-284 / turbocoder.c:230-255
    const uint32_t L  = (n + 63u) / 64u;
    const uint32_t i0 = min(lane * L, n), i1 = min(i0 + L, n);
    uint32_t       v  = 0;
    for (uint32_t i = i0; i < i1; i++) {
      v = crc24_bit(v, sm[3u * i], CRC24B_POLY);
    }
    v = wave_xor(mulmod24(v, p.crc_mult[job.crc_mult_row * 64u + lane], CRC24B_POLY));
    if (lane < 24) {
      sm[3u * (n + lane)] = (v >> (23u - lane)) & 1u;
    }
    __syncthreads();
  }
  encode_block(sm, K, job.f1, job.f2);
  __syncthreads();
  // rate matching (rm_turbo.c:340-378, TS 36.212 5.1.4.1.2): output bit k = d[table[k mod Ncb]], d natural = 3 i + stream;
  // the code blocks are concatenated bit-wise (E need not be a multiple of 8): whole bytes are stored, the two partial bytes
  // at the ends are OR-ed in atomically (the output is zeroed beforehand)
  const uint32_t first = job.out_bit, last = job.out_bit + job.E; // [first, last)
  for (uint32_t byte = (first >> 3) + lane; byte <= ((last - 1) >> 3) && job.E; byte += 64) {
    uint32_t v = 0, mask = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t ob = byte * 8u + k;
      if (ob >= first && ob < last) {
        const uint32_t bit = sm[job.table[(ob - first) % job.table_len]];
        v |= bit << (7u - k);
        mask |= 1u << (7u - k);
      }
    }
    if (mask == 0xffu) {
      p.e_bits[byte] = (uint8_t)v;
    } else {
      atomicOr((unsigned int*)(p.e_bits + (byte & ~3u)), v << (8u * (byte & 3u)));
    }
  }
}

// ---- latency kernel: one workgroup of 256 lanes per code block, everything of sch.c:238-345 for that block in one launch ----------------------
//
// The throughput kernel above gives a code block to ONE wave: 96 trellis steps per lane and encoder, twice (once from the zero state to learn the
// lanes' exit states, once more from the true entry state), behind a 63-step serial hand-over of the entry states -- 55 us for a transport block,
// whatever its size, next to a separate CRC24A launch.  Here a block has 256 lanes (24 steps each at K = 6144), every encoder runs ONCE from the zero
// state keeping its parity bits in a register, the entry states come from a logarithmic scan of the lanes' affine maps (state' = A^len state + v;
// A has period 7), and because the encoder is linear the true parity bits are the zero-state ones XOR the zero-input response of the entry
// state (period 7).  The transport-block CRC is formed by the workgroup of the block that carries it.

// Zero-input steps of the constituent encoder are linear in the state: A^e s (e = 0 ... 6: A has period 7) and the parity bits the next steps put out
// from state s are XORs of what the three unit states give -- compile-time constants picked by the bits of s.  (As tables indexed by e / s the
// compiler kept them in SCRATCH memory: a dependent scratch load per step of the scan.)
__host__ __device__ constexpr uint32_t rsc_next0(uint32_t s) // rsc_step(s, 0)
{
  const uint32_t r0 = s & 1u, r1 = (s >> 1) & 1u, r2 = (s >> 2) & 1u;
  return (r2 ^ r1) | (r0 << 1) | (r1 << 2);
}
__host__ __device__ constexpr uint32_t rsc_out0(uint32_t s) // its parity output: r2 ^ r0 ^ (r2 ^ r1)
{
  return (s ^ (s >> 1)) & 1u;
}
__host__ __device__ constexpr uint32_t rsc_pow_col(int j) // A^e (1 << j) at bits 3 e, e = 0 ... 6
{
  uint32_t s = 1u << j, w = 0;
  for (int e = 0; e < 7; e++) {
    w |= s << (3 * e);
    s = rsc_next0(s);
  }
  return w;
}
__host__ __device__ constexpr uint32_t rsc_resp_col(int j) // bit i = parity output of zero-input step i from state 1 << j, period 7, up to bit 27
{
  uint32_t s = 1u << j, r = 0;
  for (int i = 0; i < 7; i++) {
    r |= rsc_out0(s) << i;
    s = rsc_next0(s);
  }
  return r | (r << 7) | (r << 14) | (r << 21);
}
static_assert(rsc_next0(rsc_next0(rsc_next0(rsc_next0(rsc_next0(rsc_next0(rsc_next0(5u))))))) == 5u, "period 7");
__device__ __forceinline__ uint32_t rsc_resp(uint32_t s)
{
  constexpr uint32_t R0 = rsc_resp_col(0), R1 = rsc_resp_col(1), R2 = rsc_resp_col(2);
  return ((s & 1u) ? R0 : 0u) ^ ((s & 2u) ? R1 : 0u) ^ ((s & 4u) ? R2 : 0u);
}
__device__ __forceinline__ uint32_t rsc_apply(uint32_t e, uint32_t s) // A^e s, e < 7
{
  constexpr uint32_t P0 = rsc_pow_col(0), P1 = rsc_pow_col(1), P2 = rsc_pow_col(2);
  const uint32_t     sh = 3u * e;
  return (((s & 1u) ? (P0 >> sh) : 0u) ^ ((s & 2u) ? (P1 >> sh) : 0u) ^ ((s & 4u) ? (P2 >> sh) : 0u)) & 7u;
}

// XOR over the 256 lanes of the workgroup (4 waves); red: 4 words of LDS
__device__ __forceinline__ uint32_t wg_xor(uint32_t v, uint32_t* red)
{
  v = wave_xor(v);
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) {
    red[threadIdx.x >> 6] = v;
  }
  __syncthreads();
  return red[0] ^ red[1] ^ red[2] ^ red[3];
}

__global__ __launch_bounds__(256) void tb_encode_lat_kernel(const TbParams p)
{
  extern __shared__ uint8_t sm[]; // the natural code word d (bit per byte), 3 K + 12
  __shared__ uint32_t       red[4];
  __shared__ uint32_t       wtot[2][4]; // per encoder: the four waves' (exponent << 3 | v) maps
  const TbCbJob  job  = p.cbs[blockIdx.x];
  const uint32_t K    = job.K;
  const uint32_t lane = threadIdx.x;
  // ---- payload bits of this block, a dword of payload per lane and pass
  {
    const uint32_t b0 = job.src_bit, nb = job.n_src_bits;
    const uint32_t w0 = b0 >> 5; // first payload word that holds a bit of this block
    const uint32_t nw = ((b0 + nb + 31u) >> 5) - w0;
    const uint32_t* dw    = reinterpret_cast<const uint32_t*>(p.data);
    const uint32_t  lastb = (b0 + nb - 1u) >> 3; // last payload byte of this block: nothing behind it is read
    for (uint32_t w = lane; w < nw; w += 256) {
      uint32_t v; // payload bytes are MSB first: bit k of the word (from the left) = bit 31 - k
      if (4u * (w0 + w) + 3u <= lastb) {
        v = __builtin_bswap32(dw[w0 + w]);
      } else {
        v = 0;
        for (uint32_t b = 4u * (w0 + w), sh = 24; b <= lastb; b++, sh -= 8) {
          v |= (uint32_t)p.data[b] << sh;
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < 32; k++) {
        const uint32_t gb = (w0 + w) * 32u + k;
        if (gb >= b0 && gb < b0 + nb) {
          sm[3u * (gb - b0)] = (uint8_t)((v >> (31u - k)) & 1u);
        }
      }
    }
  }
  uint32_t n = job.n_src_bits;
  if (job.tb_crc != 0xffffffffu) { // the transport-block CRC closes the last block (sch.c:268-273): this workgroup forms it over the whole payload
    const TbCrcJob  tj = p.tbs[job.tb_crc];
    const uint8_t*  d  = p.data + tj.src_byte;
    const uint32_t  L  = (tj.n_bytes + 255u) / 256u;
    const int64_t   e1 = (int64_t)tj.n_bytes - (int64_t)(255u - lane) * L; // end of this lane's stretch (exclusive)
    const int64_t   e0 = e1 - (int64_t)L;
    uint32_t        c  = 0;
    for (int64_t i = e0 < 0 ? 0 : e0; i < e1; i++) {
      const uint32_t b = d[i];
#pragma unroll
      for (int k = 7; k >= 0; k--) {
        c = crc24_bit(c, (b >> k) & 1u, CRC24A_POLY);
      }
    }
    c = wg_xor(mulmod24(c, p.crc_mult256[tj.crc_mult256_row * 256u + lane], CRC24A_POLY), red);
    if (lane < 24) {
      sm[3u * (n + lane)] = (c >> (23u - lane)) & 1u;
    }
    n += 24;
  }
  __syncthreads();
  if (job.crc24b) { // sch.c:276-284 / turbocoder.c:230-255
    const uint32_t L  = (n + 255u) / 256u;
    const int32_t  e1 = (int32_t)n - (int32_t)((255u - lane) * L), e0 = e1 - (int32_t)L;
    uint32_t       v  = 0;
    for (int32_t i = e0 < 0 ? 0 : e0; i < e1; i++) {
      v = crc24_bit(v, sm[3u * (uint32_t)i], CRC24B_POLY);
    }
    v = wg_xor(mulmod24(v, p.crc_mult256[job.crc_mult256_row * 256u + lane], CRC24B_POLY), red);
    if (lane < 24) {
      sm[3u * (n + lane)] = (v >> (23u - lane)) & 1u;
    }
    __syncthreads();
  }
  // ---- both constituent encoders from the zero state: parity bits of this lane's stretch in a register (L <= 24 steps)
  const uint32_t  L  = (K + 255u) / 256u;
  const uint32_t  i0 = min(lane * L, K), i1 = min(i0 + L, K), len = i1 - i0;
  uint32_t        par[2], fin[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    uint32_t s = 0, o, bits = 0;
    Qpp      q;
    q.start(i0, K, job.f1, job.f2);
    for (uint32_t i = i0, j = 0; i < i1; i++, j++) {
      const uint32_t b = sm[3u * (e == 0 ? i : q.pi)];
      s                = rsc_step(s, b, &o);
      bits |= o << j;
      q.next();
    }
    // this lane's map: state' = A^(len mod 7) state + s.  Inclusive scan over the lanes of the wave (maps compose: later after earlier) ...
    uint32_t me = len % 7u, mv = s;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
      const uint32_t pe = __shfl_up(me, off), pv = __shfl_up(mv, off);
      if ((lane & 63u) >= off) {
        mv = rsc_apply(me, pv) ^ mv; // (A^me (A^pe x + pv) + mv
        me = (me + pe) % 7u;
      }
    }
    // ... then over the four waves
    if ((lane & 63u) == 63u) {
      wtot[e][lane >> 6] = (me << 3) | mv;
    }
    __syncthreads();
    uint32_t s_wave = 0; // state at the start of this lane's wave
    for (uint32_t w = 0; w < (lane >> 6); w++) {
      s_wave = rsc_apply(wtot[e][w] >> 3, s_wave) ^ (wtot[e][w] & 7u);
    }
    // entry state of the lane = the maps of all lanes before it applied to the zero state: the exclusive prefix
    const uint32_t pe = __shfl_up(me, 1), pv = __shfl_up(mv, 1);
    const uint32_t s_in = (lane & 63u) ? (rsc_apply(pe, s_wave) ^ pv) : s_wave;
    // the encoder is linear: true parity = zero-state parity + zero-input response of the entry state
    par[e] = bits ^ (rsc_resp(s_in) & ((1u << len) - 1u)); // (len <= 24)
    // final state of the whole block (for the tail): what lane 255 leaves
    uint32_t s_end = s_wave;
    for (uint32_t w = (lane >> 6); w < 4; w++) {
      s_end = rsc_apply(wtot[e][w] >> 3, s_end) ^ (wtot[e][w] & 7u);
    }
    fin[e] = s_end;
    __syncthreads();
  }
  for (uint32_t j = 0; j < len; j++) {
    sm[3u * (i0 + j) + 1] = (uint8_t)((par[0] >> j) & 1u);
    sm[3u * (i0 + j) + 2] = (uint8_t)((par[1] >> j) & 1u);
  }
  if (lane < 2) { // turbocoder.c:150-185: three (systematic, parity) pairs per constituent encoder
    uint32_t s = fin[lane];
    for (int j = 0; j < 3; j++) {
      const uint32_t bit = ((s >> 2) ^ (s >> 1)) & 1u;
      uint32_t       o;
      s                              = rsc_step(s, bit, &o);
      sm[3u * K + lane * 6 + 2 * j]     = (uint8_t)bit;
      sm[3u * K + lane * 6 + 2 * j + 1] = (uint8_t)o;
    }
  }
  __syncthreads();
  // ---- rate matching, as in the throughput kernel
  const uint32_t first = job.out_bit, last = job.out_bit + job.E; // [first, last)
  for (uint32_t byte = (first >> 3) + lane; byte <= ((last - 1) >> 3) && job.E; byte += 256) {
    uint32_t v = 0, mask = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t ob = byte * 8u + k;
      if (ob >= first && ob < last) {
        uint32_t idx = ob - first;
        idx -= idx >= job.table_len ? (idx / job.table_len) * job.table_len : 0u;
        const uint32_t bit = sm[job.table[idx]];
        v |= bit << (7u - k);
        mask |= 1u << (7u - k);
      }
    }
    if (mask == 0xffu) {
      p.e_bits[byte] = (uint8_t)v;
    } else {
      atomicOr((unsigned int*)(p.e_bits + (byte & ~3u)), v << (8u * (byte & 3u)));
    }
  }
}

// ---- byte-packed per-block entry points (srsran_tcod_encode_lut / srsran_rm_turbo_tx_lut, turbocoder.c:188-343, rm_turbo.c:340-378)
// One code block per launch: these exist for link-level compatibility; the batched transport-block kernel above is the
// throughput path.

__device__ __forceinline__ uint32_t wave_crc24(const uint8_t* d, uint32_t n, uint32_t poly) // remainder of d[3 i], i < n (zero start)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t L    = (n + 63u) / 64u;
  const uint32_t i0 = min(lane * L, n), i1 = min(i0 + L, n);
  uint32_t       v = 0;
  for (uint32_t i = i0; i < i1; i++) {
    v = crc24_bit(v, d[3u * i], poly);
  }
  return wave_xor(mulmod24(v, xpow24(n - i1, poly), poly));
}

__global__ __launch_bounds__(64) void lut_encode_kernel(const LutEncParams p)
{
  extern __shared__ uint8_t sm[]; // the natural code word d
  const uint32_t K = p.K, lane = threadIdx.x;
  uint32_t       n = 8u * p.n_data_bytes;
  for (uint32_t i = lane; i < n; i += 64) {
    sm[3u * i] = (p.in[i >> 3] >> (7u - (i & 7u))) & 1u;
  }
  __syncthreads();
  // running transport-block checksum over the data bytes of this block (turbocoder.c:222-228,268-272)
  const uint32_t tb_poly = p.tb_poly & 0xffffffu, cb_poly = p.cb_poly & 0xffffffu;
  uint32_t       tb      = wave_crc24(sm, n, tb_poly) ^ mulmod24(p.crc_state[0] & 0xffffffu, xpow24(n, tb_poly), tb_poly);
  if (p.last_cb) { // :233-247 / :275-287: the transport-block CRC closes the last block
    if (lane < 24) {
      sm[3u * (n + lane)] = (tb >> (23u - lane)) & 1u;
    }
    n += 24;
    __syncthreads();
  }
  uint32_t cb = 0;
  if (p.has_cb_crc) { // :249-258
    cb = wave_crc24(sm, n, cb_poly);
    if (lane < 24) {
      sm[3u * (n + lane)] = (cb >> (23u - lane)) & 1u;
    }
    n += 24;
    __syncthreads();
  }
  encode_block(sm, K, p.f1, p.f2);
  __syncthreads();
  // systematic bytes + the systematic tail nibble (:338)
  for (uint32_t b = lane; b <= K / 8; b += 64) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t i = b * 8 + k;
      const uint32_t bit = i < K ? sm[3u * i] : (i < K + 4 ? sm[3u * K + 3u * (i - K)] : 0u);
      v |= bit << (7u - k);
    }
    p.out_sys[b] = (uint8_t)v;
  }
  // parity bit array [p1 (K) | p1 tail (4) | p2 (K) | p2 tail (4)] (:212-216,294-303,340-342)
  for (uint32_t b = lane; b < (2 * K + 8) / 8; b += 64) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t q = b * 8 + k;
      uint32_t       bit;
      if (q < K) {
        bit = sm[3u * q + 1];
      } else if (q < K + 4) {
        bit = sm[3u * K + 3u * (q - K) + 1];
      } else if (q < 2 * K + 4) {
        bit = sm[3u * (q - K - 4) + 2];
      } else {
        bit = sm[3u * K + 3u * (q - 2 * K - 4) + 2];
      }
      v |= bit << (7u - k);
    }
    p.out_par[b] = (uint8_t)v;
  }
  if (lane == 0) {
    p.crc_state[0] = tb;
    p.crc_state[1] = cb;
  }
}

// E rate-matched bits (packed from bit 0) of one block given its byte-packed streams
__global__ __launch_bounds__(256) void lut_rm_kernel(const LutRmParams p)
{
  const uint32_t K = p.K;
  auto           bit_of = [&](const uint8_t* a, uint32_t i) { return (uint32_t)(a[i >> 3] >> (7u - (i & 7u))) & 1u; };
  for (uint32_t byte = threadIdx.x; byte < (p.E + 7) / 8; byte += 256) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < 8 && byte * 8 + k < p.E; k++) {
      const uint32_t t = p.table[(byte * 8 + k) % p.table_len];
      uint32_t       i, s;
      if (t < 3 * K) {
        i = t / 3u;
        s = t - 3u * i;
      } else {
        i = K + (t - 3 * K) / 3u;
        s = (t - 3 * K) % 3u;
      }
      const uint32_t bit = s == 0 ? bit_of(p.sys, i) : (s == 1 ? bit_of(p.par, i) : bit_of(p.par, K + 4 + i));
      v |= bit << (7u - k);
    }
    p.out[byte] = (uint8_t)v;
  }
}

} // namespace

hipError_t launch_encode(const EncParams& p, hipStream_t stream)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(encode_kernel, dim3(p.n_cb), dim3(64), 3 * p.K + 16, stream, p);
  return hipGetLastError();
}

hipError_t launch_lut_encode(const LutEncParams& p, hipStream_t stream)
{
  hipLaunchKernelGGL(lut_encode_kernel, dim3(1), dim3(64), 3 * p.K + 16, stream, p);
  return hipGetLastError();
}

hipError_t launch_lut_rm(const LutRmParams& p, hipStream_t stream)
{
  if (p.E == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(lut_rm_kernel, dim3(1), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_tb_crc24a(const TbParams& p, hipStream_t stream)
{
  if (p.n_tb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(tb_crc24a_kernel, dim3(p.n_tb), dim3(64), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_tb_encode_lat(const TbParams& p, hipStream_t stream)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(tb_encode_lat_kernel, dim3(p.n_cb), dim3(256), 3 * 6144 + 16, stream, p);
  return hipGetLastError();
}

hipError_t launch_tb_encode(const TbParams& p, hipStream_t stream)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(tb_encode_kernel, dim3(p.n_cb), dim3(64), 3 * 6144 + 16, stream, p);
  return hipGetLastError();
}

} // namespace tcod
} // namespace phyhip
