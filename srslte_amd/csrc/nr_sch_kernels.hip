// nr_sch_kernels.hip -- NR shared-channel bit processing around the LDPC decoder (gfx950):
//   rm_rx_kernel   srsran_ldpc_rm_rx_{c,s,f}  (ldpc_rm.c:203-346,365-411): de-interleave + de-select + HARQ accumulate
//   rm_tx_kernel   srsran_ldpc_rm_tx          (ldpc_rm.c:173-193,348-362): select + interleave
//   encode_kernel  srsran_ldpc_encoder_encode_rm (ldpc_encoder.c:55-95, ldpc_enc_c.c)
//
// The reference's de-matcher is a sequential scatter-accumulate in transmission order (the saturation after every addition
// makes the order matter when a position is received more than once).  Here the rank -> position map of the circular,
// filler-skipping read-out order is closed-form, one lap of the buffer touches every position at most once, and laps are
// separate launches: no atomics, no temporary buffer, no index table.
#include "hip_common.h"
#include "nr_sch_device.h"

namespace phyhip {
namespace nrsch {

namespace {

template <typename T>
struct Acc;
template <>
struct Acc<int8_t> { // ldpc_rm.c:323-345
  static __device__ __forceinline__ int8_t inf() { return 127; }
  static __device__ __forceinline__ int8_t add(int8_t a, int8_t b) { return (int8_t)min(max((int)a + (int)b, -63), 63); }
};
template <>
struct Acc<int16_t> { // ldpc_rm.c:270-291
  static __device__ __forceinline__ int16_t inf() { return 32767; }
  static __device__ __forceinline__ int16_t add(int16_t a, int16_t b) { return (int16_t)min(max((int)a + (int)b, -16383), 16383); }
};
template <>
struct Acc<float> { // ldpc_rm.c:226-234
  static __device__ __forceinline__ float inf() { return __builtin_inff(); }
  static __device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
};

// fillers among the positions [0, x)
__device__ __forceinline__ uint32_t fcount(const RmParams& p, uint32_t x)
{
  return min(max(x, p.ini_ex), p.end_ex) - p.ini_ex;
}

// k / cols and k % cols for k < 2^24 through a float reciprocal (one correction step)
__device__ __forceinline__ void divmod(uint32_t k, uint32_t cols, float inv, uint32_t* q, uint32_t* r)
{
  uint32_t qq = (uint32_t)__float2uint_rz(__uint2float_rn(k) * inv);
  int      rr = (int)(k - qq * cols);
  if (rr < 0) {
    qq--;
    rr += (int)cols;
  } else if ((uint32_t)rr >= cols) {
    qq++;
    rr -= (int)cols;
  }
  *q = qq;
  *r = (uint32_t)rr;
}

// Position of rank k (k < P) in the read-out order that starts at k0, wraps at Ncb and skips the filler range [fi, fe).
__device__ __forceinline__ uint32_t rx_position(const RmParams& p, uint32_t fi, uint32_t fe, uint32_t A, uint32_t k)
{
  uint32_t s = p.k0, q = k;
  if (k >= A) { // second stretch: from the start of the buffer
    s = 0;
    q = k - A;
  } else if (s >= fi && s < fe) { // k0 inside the fillers: the read-out really starts behind them
    s = fe;
  }
  uint32_t pos = s + q;
  if (s <= fi && pos >= fi) {
    pos += fe - fi;
  }
  return pos;
}

template <typename T>
__global__ __launch_bounds__(256) void rm_rx_kernel(const RmParams p, uint32_t lap)
{
  // Driven by the INPUT: a lane takes one modulation symbol j, i.e. the Qm consecutive soft bits in[j Qm .. j Qm + Qm - 1], which
  // the de-interleaver sends to the ranks i * cols + j of the read-out order (ldpc_rm.c:365-411).  For a fixed i consecutive
  // lanes hold consecutive ranks, hence consecutive soft-buffer positions: loads and read-modify-writes are coalesced without
  // any index arithmetic beyond the closed-form rank -> position map.  Within one lap of the circular buffer (P ranks) every
  // position receives at most one soft bit, so the lanes never collide; further laps (repetition, E > P) are separate
  // launches in stream order, which also keeps the reference's saturation order.  The workgroups behind the symbol tiles
  // write the filler "infinities" (first lap only).
  const CbJob    job = p.jobs[blockIdx.y];
  const T*       in  = (const T*)p.in + job.in_off;
  T*             out = (T*)p.out + job.out_off;
  const uint32_t E = job.E, cols = E / p.Qm;
  const uint32_t fi = min(p.ini_ex, p.Ncb), fe = min(p.end_ex, p.Ncb);
  const uint32_t P = p.Ncb - (fe - fi);
  const uint32_t A = (p.Ncb - p.k0) - (fe - min(max(p.k0, fi), fe)); // non-filler positions in [k0, Ncb)
  const uint32_t spl = p.Qm >= 8 ? 1u : 8u / p.Qm; // symbols per lane: about 8 soft bits whatever the modulation
  const uint32_t sym_tiles = (cols + 256u * spl - 1) / (256u * spl);
  const bool fresh = lap == 0 && (job.aux & 1u); // new data: nothing to accumulate into, every position of the lap is written
  if (blockIdx.x >= sym_tiles) {
    const uint32_t t = blockIdx.x - sym_tiles, fill_tiles = (p.end_ex - p.ini_ex + 255u) / 256u;
    if (t < fill_tiles) {
      const uint32_t i = p.ini_ex + t * 256u + threadIdx.x;
      if (lap == 0 && i < p.end_ex) {
        out[i] = Acc<T>::inf(); // ldpc_rm.c:226-229,266-269,318-321
      }
    } else if (fresh && P) {
      // the ranks this transmission does not reach: what a cleared soft buffer holds there
      for (uint32_t u = 0; u < 8; u++) { // 2048 ranks per tile
        const uint32_t k = min(E, P) + ((t - fill_tiles) * 8u + u) * 256u + threadIdx.x;
        if (k < P) {
          out[rx_position(p, fi, fe, A, k)] = (T)0;
        }
      }
    }
    return;
  }
  if (P == 0) {
    return;
  }
  const uint32_t lo = lap * P, hi = min(E, lo + P); // ranks of this lap
  for (uint32_t s = 0; s < spl; s++) {
    const uint32_t j = (blockIdx.x * spl + s) * 256u + threadIdx.x;
    if (j >= cols) {
      break;
    }
    for (uint32_t i = 0; i < p.Qm; i++) {
      const uint32_t k = i * cols + j;
      if (k >= lo && k < hi) {
        const uint32_t pos = rx_position(p, fi, fe, A, k - lo);
        out[pos]           = Acc<T>::add(fresh ? (T)0 : out[pos], in[j * p.Qm + i]);
      }
    }
  }
}

// ---- transmit side: one workgroup per code block ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rm_tx_kernel(const RmParams p)
{
  extern __shared__ uint16_t pos_of_rank[]; // Ncb entries
  __shared__ uint32_t        part[257];
  const CbJob                job = p.jobs[blockIdx.x];
  const uint8_t*             in  = (const uint8_t*)p.in + job.in_off;
  uint8_t*                   out = (uint8_t*)p.out + job.out_off;
  // read-out order: circular offsets from k0; every lane counts the non-filler bits of its stretch (ldpc_rm.c:186-191
  // tests the VALUE of the bit, wherever it is)
  const uint32_t c  = (p.Ncb + 255u) / 256u;
  const uint32_t o0 = min(threadIdx.x * c, p.Ncb), o1 = min(o0 + c, p.Ncb);
  uint32_t       cnt = 0;
  for (uint32_t o = o0; o < o1; o++) {
    uint32_t q = p.k0 + o;
    q -= q >= p.Ncb ? p.Ncb : 0;
    cnt += in[q] != 254;
  }
  part[threadIdx.x + 1] = cnt;
  if (threadIdx.x == 0) {
    part[0] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i <= 256; i++) {
      part[i] += part[i - 1];
    }
  }
  __syncthreads();
  uint32_t r = part[threadIdx.x];
  for (uint32_t o = o0; o < o1; o++) {
    uint32_t q = p.k0 + o;
    q -= q >= p.Ncb ? p.Ncb : 0;
    if (in[q] != 254) {
      pos_of_rank[r++] = (uint16_t)q;
    }
  }
  const uint32_t P = part[256];
  __syncthreads();
  if (P == 0) {
    return;
  }
  const uint32_t cols = job.E / p.Qm;
  for (uint32_t o = threadIdx.x; o < job.E; o += 256) {
    const uint32_t i = o % p.Qm, j = o / p.Qm; // ldpc_rm.c:348-362: out[i + j * Qm] = tmp[i * cols + j]
    const uint32_t k = p.Qm == 1 ? o : i * cols + j;
    out[o]           = in[pos_of_rank[k % P]];
  }
}

// ---- encoder: one workgroup per code block ------------------------------------------------------------------------------------
__device__ __forceinline__ int rot(int i, int sh, int Z) // (i + sh) mod Z for 0 <= i, sh < Z
{
  const int t = i + sh;
  return t >= Z ? t - Z : t;
}

__global__ __launch_bounds__(256) void encode_kernel(const EncParams p)
{
  extern __shared__ uint8_t sm[]; // message bits (bgK Z) | lambda (4 Z) | core parity (4 Z)
  const int      Z = p.Z, bgK = p.bgK;
  uint8_t*       msg = sm;
  uint8_t*       lam = sm + bgK * Z;
  uint8_t*       par = lam + 4 * Z;
  const CbJob    job = p.jobs[blockIdx.x];
  const uint8_t* in  = p.in + job.in_off;
  uint8_t*       out = p.out + job.out_off;
  const int      n_layers = (int)job.aux;
  // systematic part: raw copy without the two punctured blocks (ldpc_encoder.c:80-84); filler flags count as 0 below
  for (int i = threadIdx.x; i < bgK * Z; i += 256) {
    const uint8_t b = in[i];
    msg[i]          = b & 1;
    if (i >= 2 * Z) {
      out[i - 2 * Z] = b;
    }
  }
  __syncthreads();
  // lambda_m = systematic contribution to core row m
  for (int t = threadIdx.x; t < 4 * Z; t += 256) {
    const int m = t / Z, i = t - m * Z;
    uint8_t   v = 0;
    for (int e = p.row_start[m]; e < p.row_start[m + 1]; e++) {
      const int ed = p.edges[e], col = ed & 0xff;
      if (col < bgK) {
        v ^= msg[col * Z + rot(i, ed >> 8, Z)];
      }
    }
    lam[t] = v;
  }
  __syncthreads();
  // first core parity block: the four core rows add up to one rotation of it
  for (int k = threadIdx.x; k < Z; k += 256) {
    par[rot(k, p.a, Z)] = lam[k] ^ lam[Z + k] ^ lam[2 * Z + k] ^ lam[3 * Z + k];
  }
  __syncthreads();
  // the other three, each from a core row in which it is the only unknown
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const EncStep& st = p.step[s];
    for (int k = threadIdx.x; k < Z; k += 256) {
      uint8_t v = lam[st.row * Z + k];
      for (int t = 0; t < st.n_terms; t++) {
        v ^= par[st.blk[t] * Z + rot(k, st.sh[t], Z)];
      }
      par[st.unk * Z + rot(k, st.ush, Z)] = v;
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < 4 * Z; t += 256) {
    out[(bgK - 2) * Z + t] = par[t];
  }
  // extension rows: their own parity block has shift 0, everything else is known (ldpc_enc_c.c:30-62)
  for (int t = threadIdx.x; t < (n_layers - 4) * Z; t += 256) {
    const int m = 4 + t / Z, i = t - (m - 4) * Z;
    uint8_t   v = 0;
    for (int e = p.row_start[m]; e < p.row_start[m + 1]; e++) {
      const int ed = p.edges[e], col = ed & 0xff;
      if (col < bgK + 4) {
        const int idx = rot(i, ed >> 8, Z);
        v ^= col < bgK ? msg[col * Z + idx] : par[(col - bgK) * Z + idx];
      }
    }
    out[(bgK + m - 2) * Z + i] = v;
  }
}

} // namespace

hipError_t launch_rm_rx(const RmParams& p, uint32_t max_E, hipStream_t stream, uint32_t min_E_new)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  // symbol tiles (the longest code block decides; shorter ones leave early) + tiles that write the filler positions;
  // one launch per lap of the circular buffer
  const uint32_t fi = p.ini_ex < p.Ncb ? p.ini_ex : p.Ncb, fe = p.end_ex < p.Ncb ? p.end_ex : p.Ncb;
  const uint32_t P  = p.Ncb - (fe - fi);
  const uint32_t laps = P ? ceil_div(max_E ? max_E : 1u, P) : 1u;
  const uint32_t spl = p.Qm >= 8 ? 1u : 8u / p.Qm;
  // + tiles that clear the ranks a new block's first transmission does not reach (min_E_new: its shortest length, ~0u: no new block)
  const uint32_t zero_tiles = (min_E_new != ~0u && min_E_new < P) ? ceil_div(P - min_E_new, 2048u) : 0u;
  dim3           grid(ceil_div(ceil_div(max_E, p.Qm), 256u * spl) + ceil_div(p.end_ex - p.ini_ex, 256u) + zero_tiles, p.n_cb);
  for (uint32_t lap = 0; lap < laps; lap++) {
    switch (p.type) {
      case T_I8:
        hipLaunchKernelGGL(rm_rx_kernel<int8_t>, grid, dim3(256), 0, stream, p, lap);
        break;
      case T_I16:
        hipLaunchKernelGGL(rm_rx_kernel<int16_t>, grid, dim3(256), 0, stream, p, lap);
        break;
      default:
        hipLaunchKernelGGL(rm_rx_kernel<float>, grid, dim3(256), 0, stream, p, lap);
        break;
    }
  }
  return hipGetLastError();
}

hipError_t launch_rm_tx(const RmParams& p, hipStream_t stream)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(rm_tx_kernel, dim3(p.n_cb), dim3(256), p.Ncb * sizeof(uint16_t), stream, p);
  return hipGetLastError();
}

hipError_t launch_encode(const EncParams& p, hipStream_t stream)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(encode_kernel, dim3(p.n_cb), dim3(256), (size_t)(p.bgK + 8) * p.Z, stream, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ transport-block loop
// sch_nr.c:633-652: a code block counts as decoded when the CRC stopped the decoder AND its bits are not all zero; its bits are
// then packed MSB first.  One workgroup per code block.
__global__ __launch_bounds__(256) void cb_finish_kernel(const uint8_t* msg, uint32_t msg_stride, const CbFin* jobs, const int* n_iter, uint8_t* flags,
                                                        uint8_t* cb_data, uint32_t data_stride)
{
  const CbFin    jb   = jobs[blockIdx.x];
  const uint8_t* bits = msg + (size_t)jb.msg_row * msg_stride;
  const uint32_t nb   = (jb.cb_len + 7) / 8;
  int            any  = 0;
  for (uint32_t b = threadIdx.x; b < nb; b += 256) {
    for (uint32_t k = 0; k < 8 && b * 8 + k < jb.cb_len; k++) {
      any |= bits[b * 8 + k];
    }
  }
  const bool ok = __syncthreads_or(any) != 0 && n_iter[blockIdx.x] != 0;
  if (threadIdx.x == 0) {
    flags[jb.cb_index] = ok ? 1 : 0;
  }
  if (!ok) {
    return;
  }
  uint8_t* d = cb_data + (size_t)jb.cb_index * data_stride;
  for (uint32_t b = threadIdx.x; b < nb; b += 256) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < 8 && b * 8 + k < jb.cb_len; k++) {
      v |= (uint32_t)(bits[b * 8 + k] & 1u) << (7 - k);
    }
    d[b] = (uint8_t)v;
  }
}

__device__ __forceinline__ uint32_t gf_mulmod_n(uint32_t a, uint32_t b, uint32_t poly, uint32_t order)
{
  const uint32_t mask = (1u << order) - 1u;
  uint32_t       r    = 0;
  for (int i = (int)order - 1; i >= 0; i--) {
    r = ((r << 1) & mask) ^ (((r >> (order - 1)) & 1u) ? poly : 0u);
    r ^= ((b >> i) & 1u) ? a : 0u;
  }
  return r;
}

// sch_nr.c:667-705: when every code block of a transport block is decoded, append their payload parts, take the transport CRC
// carried behind the payload in the last block and compare it with the CRC of the payload (CRC24A above 3824 bits, else CRC16;
// crc.c, MSB first, zero initial state).  One workgroup per transport block.  The payload, padded in front with zero bytes to 256 equal
// chunks, is split over the lanes: a byte per look-up in a 256-entry table of the generator (built in LDS), then one multiplication by
// x^(bits behind the chunk) mod g from the host's table for this chunk size (tb_finish_multipliers) -- the per-lane square-and-multiply
// that used to produce that power was most of the kernel (0.092 -> 0.03 ms per 1024 blocks of 8).
__global__ __launch_bounds__(256) void tb_finish_kernel(const uint8_t* cb_data, uint32_t data_stride, const uint8_t* flags, const TbFin* jobs,
                                                        uint8_t* payload, const uint32_t* mult, TbFinRes* res)
{
  __shared__ uint32_t red[256];
  __shared__ uint32_t tab[256];
  const TbFin jb  = jobs[blockIdx.x];
  int         bad = 0;
  for (uint32_t r = threadIdx.x; r < jb.C; r += 256) {
    bad |= flags[jb.first_cb + r] ? 0 : 1;
  }
  if (__syncthreads_or(bad)) {
    if (threadIdx.x == 0) {
      res[blockIdx.x].all_decoded = 0;
      res[blockIdx.x].crc_ok      = 0;
    }
    return;
  }
  const uint32_t per = (jb.Kp - jb.L_cb) / 8; // payload bytes of every code block but the last
  const uint32_t nb  = jb.tbs / 8;
  auto byte_at = [&](uint32_t i) {
    const uint32_t r = i / per;
    return cb_data[(size_t)(jb.first_cb + r) * data_stride + (i - r * per)];
  };
  const uint32_t order = jb.L_tb, mask = (1u << order) - 1u, poly = (order == 24 ? 0x1864CFBu : 0x11021u) & mask;
  {
    uint32_t c = (uint32_t)threadIdx.x << (order - 8);
    for (int b = 0; b < 8; b++) {
      c = ((c << 1) & mask) ^ (((c >> (order - 1)) & 1u) ? poly : 0u);
    }
    tab[threadIdx.x] = c;
  }
  __syncthreads();
  const uint32_t c   = (nb + 255) / 256, pad = 256 * c - nb; // the first `pad` bytes of the padded payload are zeros
  const uint32_t plo = threadIdx.x * c, phi = plo + c;
  uint32_t       crc = 0;
  uint8_t*       out = payload + jb.payload_off;
  for (uint32_t pi = plo < pad ? pad : plo; pi < phi; pi++) {
    const uint32_t i    = pi - pad;
    const uint32_t byte = byte_at(i);
    out[i]              = (uint8_t)byte;
    crc                 = ((crc << 8) & mask) ^ tab[((crc >> (order - 8)) ^ byte) & 0xffu];
  }
  red[threadIdx.x] = gf_mulmod_n(crc, mult[256u * jb.mult + threadIdx.x], poly, order);
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x] ^= red[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint32_t       checksum2 = 0;
    const uint32_t last      = jb.C - 1, at = nb - last * per; // byte behind the payload part of the last block
    for (uint32_t i = 0; i < order / 8; i++) {
      checksum2 = (checksum2 << 8) | cb_data[(size_t)(jb.first_cb + last) * data_stride + at + i];
    }
    res[blockIdx.x].all_decoded = 1;
    res[blockIdx.x].crc_ok      = (jb.C == 1 || red[0] == checksum2) ? 1 : 0;
  }
}

uint32_t tb_finish_chunk(uint32_t tbs_bits)
{
  return (tbs_bits / 8 + 255) / 256;
}

void tb_finish_multipliers(uint32_t chunk, uint32_t order, uint32_t out[256])
{
  const uint32_t mask = (1u << order) - 1u, poly = (order == 24 ? 0x1864CFBu : 0x11021u) & mask;
  auto           mul  = [&](uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int i = (int)order - 1; i >= 0; i--) {
      r = ((r << 1) & mask) ^ (((r >> (order - 1)) & 1u) ? poly : 0u);
      r ^= ((b >> i) & 1u) ? a : 0u;
    }
    return r;
  };
  uint32_t step = 1, base = 2 & mask, e = 8 * chunk; // x^(8 chunk) mod g
  while (e) {
    if (e & 1) {
      step = mul(step, base);
    }
    base = mul(base, base);
    e >>= 1;
  }
  out[255] = 1;
  for (int l = 254; l >= 0; l--) {
    out[l] = mul(out[l + 1], step);
  }
}

// ---- transmit side: sch_nr_encode (sch_nr.c:375-520) up to the encoder input
// CRC of `n` bits handed out by bit_at(i), MSB first, zero initial state (crc.c), by all 256 lanes of the workgroup
template <class F>
__device__ __forceinline__ uint32_t wg_crc_bits(F bit_at, uint32_t n, uint32_t poly_full, uint32_t order, uint32_t* red)
{
  const uint32_t mask = (1u << order) - 1u, poly = poly_full & mask;
  const uint32_t c  = (n + 255) / 256;
  const uint32_t lo = threadIdx.x * c, hi = lo + c < n ? lo + c : n;
  uint32_t       crc = 0;
  for (uint32_t i = lo; i < hi; i++) {
    crc = ((crc << 1) & mask) ^ ((((crc >> (order - 1)) ^ bit_at(i)) & 1u) ? poly : 0u);
  }
  if (lo < n) {
    uint32_t e = n - hi, result = 1, base = 2; // x^e mod g
    while (e) {
      if (e & 1) {
        result = gf_mulmod_n(result, base, poly, order);
      }
      base = gf_mulmod_n(base, base, poly, order);
      e >>= 1;
    }
    crc = gf_mulmod_n(crc, result, poly, order);
  } else {
    crc = 0;
  }
  __syncthreads();
  red[threadIdx.x] = crc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x] ^= red[threadIdx.x + s];
    }
    __syncthreads();
  }
  return red[0];
}

__global__ __launch_bounds__(256) void tb_crc_enc_kernel(const uint8_t* payload, const TbEnc* tbs, uint32_t* crc_out)
{
  __shared__ uint32_t red[256];
  const TbEnc    tb = tbs[blockIdx.x];
  const uint8_t* d  = payload + tb.payload_off;
  const uint32_t c  = wg_crc_bits([&](uint32_t i) { return (uint32_t)(d[i >> 3] >> (7 - (i & 7))) & 1u; }, tb.tbs,
                                 tb.L_tb == 24 ? 0x1864CFBu : 0x11021u, tb.L_tb, red);
  if (threadIdx.x == 0) {
    crc_out[blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(256) void cb_build_kernel(const uint8_t* payload, const CbEnc* cbs, const TbEnc* tbs, const uint32_t* tbcrc, uint8_t* msg,
                                                       uint32_t msg_stride)
{
  __shared__ uint8_t  bits[8448];
  __shared__ uint32_t red[256];
  const CbEnc    cb = cbs[blockIdx.x];
  const TbEnc    tb = tbs[cb.tb];
  const uint8_t* d  = payload + tb.payload_off;
  for (uint32_t i = threadIdx.x; i < cb.cb_len; i += 256) {
    const uint32_t b = cb.bit_off + i;
    bits[i]          = (d[b >> 3] >> (7 - (b & 7))) & 1u;
  }
  if (cb.last) { // :437-441
    const uint32_t c = tbcrc[cb.tb];
    for (uint32_t i = threadIdx.x; i < tb.L_tb; i += 256) {
      bits[cb.cb_len + i] = (c >> (tb.L_tb - 1 - i)) & 1u;
    }
  }
  __syncthreads();
  if (cb.L_cb) { // srsran_crc_attach over Kp - L_cb bits, :451-453
    const uint32_t n = cb.Kp - cb.L_cb;
    const uint32_t c = wg_crc_bits([&](uint32_t i) { return (uint32_t)bits[i]; }, n, 0x1800063u, 24, red);
    for (uint32_t i = threadIdx.x; i < 24; i += 256) {
      bits[n + i] = (c >> (23 - i)) & 1u;
    }
    __syncthreads();
  }
  uint8_t* out = msg + (size_t)cb.msg_row * msg_stride;
  for (uint32_t i = threadIdx.x; i < cb.Kr; i += 256) {
    out[i] = i < cb.Kp ? bits[i] : (uint8_t)254; // FILLER_BIT, :456-458
  }
}

hipError_t launch_tb_crc_enc(const uint8_t* d_payload, const TbEnc* d_tbs, uint32_t n_tb, uint32_t* d_crc, hipStream_t stream)
{
  if (n_tb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(tb_crc_enc_kernel, dim3(n_tb), dim3(256), 0, stream, d_payload, d_tbs, d_crc);
  return hipGetLastError();
}

hipError_t launch_cb_build(const uint8_t* d_payload, const CbEnc* d_cbs, uint32_t n_cb, const TbEnc* d_tbs, const uint32_t* d_crc, uint8_t* d_msg,
                           uint32_t msg_stride, hipStream_t stream)
{
  if (n_cb == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(cb_build_kernel, dim3(n_cb), dim3(256), 0, stream, d_payload, d_cbs, d_tbs, d_crc, d_msg, msg_stride);
  return hipGetLastError();
}

hipError_t launch_cb_finish(const uint8_t* d_msg, uint32_t msg_stride, const CbFin* d_jobs, const int* d_n_iter, uint32_t n, uint8_t* d_flags,
                            uint8_t* d_cb_data, uint32_t data_stride, hipStream_t stream)
{
  if (n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(cb_finish_kernel, dim3(n), dim3(256), 0, stream, d_msg, msg_stride, d_jobs, d_n_iter, d_flags, d_cb_data, data_stride);
  return hipGetLastError();
}

hipError_t launch_tb_finish(const uint8_t* d_cb_data, uint32_t data_stride, const uint8_t* d_flags, const TbFin* d_jobs, uint32_t n, uint8_t* d_payload,
                            const uint32_t* d_mult, TbFinRes* d_res, hipStream_t stream)
{
  if (n == 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(tb_finish_kernel, dim3(n), dim3(256), 0, stream, d_cb_data, data_stride, d_flags, d_jobs, d_payload, d_mult, d_res);
  return hipGetLastError();
}

} // namespace nrsch
} // namespace phyhip
