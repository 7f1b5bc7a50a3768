// nr_sch_kernels.hip -- NR shared-channel bit processing around the LDPC decoder (gfx950):
//   rm_rx_kernel   srsran_ldpc_rm_rx_{c,s,f}  (ldpc_rm.c:203-346,365-411): de-interleave + de-select + HARQ accumulate
//   rm_tx_kernel   srsran_ldpc_rm_tx          (ldpc_rm.c:173-193,348-362): select + interleave
//   encode_kernel  srsran_ldpc_encoder_encode_rm (ldpc_encoder.c:55-95, ldpc_enc_c.c)
//
// The reference's de-matcher is a sequential scatter-accumulate in transmission order (the saturation after every addition
// makes the order matter when a position is received more than once).  Here every lane owns positions of the soft buffer,
// works out which transmitted soft bits land on each of them -- rank of the position in the circular, filler-skipping
// read-out order, then every P-th one after it -- gathers them through the de-interleaving index and updates the position
// once.  No atomics, no temporary buffer.
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

// rank of a position in the read-out order that starts at k0, wraps at Ncb and skips the fillers; 0xffffffff: the position
// gets nothing from this transmission; 0xfffffffe: filler
__device__ __forceinline__ uint32_t rx_rank(const RmParams& p, uint32_t E, uint32_t P, uint32_t fk0, uint32_t fN, uint32_t pos)
{
  if (pos >= p.ini_ex && pos < p.end_ex) {
    return 0xfffffffeu;
  }
  if (pos >= p.Ncb || P == 0) {
    return 0xffffffffu;
  }
  const uint32_t k = pos >= p.k0 ? (pos - p.k0) - (fcount(p, pos) - fk0) : (pos + p.Ncb - p.k0) - (fN - fk0 + fcount(p, pos));
  return k < E ? k : 0xffffffffu;
}

template <typename T>
__device__ __forceinline__ T rx_accumulate(const RmParams& p, const T* in, uint32_t E, uint32_t cols, float inv, uint32_t P, uint32_t k, T cur)
{
  if (k == 0xfffffffeu) {
    return Acc<T>::inf();
  }
  for (; k < E; k += P) { // k = 0xffffffff falls through
    uint32_t src = k;
    if (p.Qm != 1) { // ldpc_rm.c:365-411: tmp[i * cols + j] = in[j * Qm + i]
      uint32_t i, j;
      divmod(k, cols, inv, &i, &j);
      src = j * p.Qm + i;
    }
    cur = Acc<T>::add(cur, in[src]);
  }
  return cur;
}

template <typename T>
__global__ __launch_bounds__(256) void rm_rx_kernel(const RmParams p)
{
  // a workgroup covers 256 V consecutive positions; lane l takes l, l + 256, ...: consecutive lanes hold consecutive ranks, so
  // the de-interleaving gathers of one instruction fall Qm elements apart (a few cache lines) instead of one line per lane
  constexpr uint32_t V   = 16 / sizeof(T);
  const CbJob        job = p.jobs[blockIdx.y];
  const T*           in  = (const T*)p.in + job.in_off;
  T*                 out = (T*)p.out + job.out_off;
  const uint32_t     E = job.E, cols = E / p.Qm;
  const float        inv = 1.0f / (float)max(cols, 1u);
  const uint32_t     fk0 = fcount(p, p.k0), fN = fcount(p, p.Ncb), P = p.Ncb - fN;
  const uint32_t     cover = max(p.Ncb, p.end_ex);
  const uint32_t     p0    = blockIdx.x * 256u * V + threadIdx.x;
  uint32_t           k[V];
  T                  cur[V];
#pragma unroll
  for (uint32_t i = 0; i < V; i++) {
    const uint32_t pos = p0 + i * 256u;
    k[i]               = pos < cover ? rx_rank(p, E, P, fk0, fN, pos) : 0xffffffffu;
    if (k[i] != 0xffffffffu) { // positions this transmission does not reach are neither read nor written
      cur[i] = out[pos];
    }
  }
#pragma unroll
  for (uint32_t i = 0; i < V; i++) {
    if (k[i] != 0xffffffffu) {
      out[p0 + i * 256u] = rx_accumulate<T>(p, in, E, cols, inv, P, k[i], cur[i]);
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

hipError_t launch_rm_rx(const RmParams& p, hipStream_t stream)
{
  if (p.n_cb == 0) {
    return hipSuccess;
  }
  const uint32_t cover = p.Ncb > p.end_ex ? p.Ncb : p.end_ex;
  const uint32_t es    = p.type == T_I8 ? 1 : (p.type == T_I16 ? 2 : 4);
  dim3           grid(ceil_div(cover, 256u * (16u / es)), p.n_cb);
  switch (p.type) {
    case T_I8:
      hipLaunchKernelGGL(rm_rx_kernel<int8_t>, grid, dim3(256), 0, stream, p);
      break;
    case T_I16:
      hipLaunchKernelGGL(rm_rx_kernel<int16_t>, grid, dim3(256), 0, stream, p);
      break;
    default:
      hipLaunchKernelGGL(rm_rx_kernel<float>, grid, dim3(256), 0, stream, p);
      break;
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

} // namespace nrsch
} // namespace phyhip
