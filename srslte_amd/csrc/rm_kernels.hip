// rm_kernels.hip -- LTE turbo rate de-matching, receive side, for gfx950.
//
// Reference behaviour: lib/src/phy/fec/turbo/rm_turbo.c:390-478 -- every received soft bit i is added (wrapping) to
// position deinter[i % out_len] of the code block's soft buffer (HARQ combining; i >= out_len = repetition).
// The table is a permutation of the buffer positions, so the sum is written gather-style: one lane per table entry
// adds up its <= ceil(E / out_len) input samples (coalesced reads) and does ONE read-modify-write: no atomics.
#include "hip_common.h"
#include "rm_device.h"

namespace phyhip {
namespace rm {

// jobs == nullptr: uniform batch, job b is `uni` with its offsets advanced by b * (in_stride, out_stride)
template <typename T>
__global__ __launch_bounds__(256) void rm_rx_kernel(const T* in, T* out, const uint16_t* tables, const RxJob* jobs, const RxJob uni,
                                                    uint32_t in_stride, uint32_t out_stride)
{
  RxJob jb;
  if (jobs) {
    jb = jobs[blockIdx.y];
  } else {
    jb = uni;
    jb.in_offset += blockIdx.y * in_stride;
    jb.out_offset += blockIdx.y * out_stride;
  }
  const uint32_t k  = blockIdx.x * 256 + threadIdx.x;
  if (k >= jb.out_len || k >= jb.in_len) {
    return;
  }
  const T* x   = in + jb.in_offset;
  int      acc = 0;
  for (uint32_t i = k; i < jb.in_len; i += jb.out_len) {
    acc += x[i];
  }
  T* dst = out + jb.out_offset + tables[jb.table + k];
  *dst   = (T)(*dst + acc);
}

// The same sum driven from the OUTPUT side: lane j owns soft-buffer position j and looks up which transmitted bit
// (if any) lands there (inverse table, 0xffff = none).  Writes are unit stride; the scattered 2-byte reads stay
// inside the code block's <= 37 KB of input, i.e. in L1/L2.  2.4 ms -> see DESIGN.md for 26,624 code blocks.
template <typename T>
__global__ __launch_bounds__(256) void rm_rx_gather_kernel(const T* in, T* out, const uint16_t* inverse, const RxJob* jobs, const RxJob uni,
                                                           uint32_t in_stride, uint32_t out_stride, uint32_t out_span)
{
  RxJob jb;
  if (jobs) {
    jb = jobs[blockIdx.y];
  } else {
    jb = uni;
    jb.in_offset += blockIdx.y * in_stride;
    jb.out_offset += blockIdx.y * out_stride;
  }
  constexpr int   V   = 16 / sizeof(T); // soft-buffer positions per lane: one 16-byte read-modify-write
  const T*        x   = in + jb.in_offset;
  T*              dst = out + jb.out_offset;
  const uint16_t* inv = inverse + jb.table;
  const uint32_t  j0  = (blockIdx.x * 256 + threadIdx.x) * V;
  if (j0 >= out_span) {
    return;
  }
  auto gather = [&](uint32_t k) {
    int acc = 0;
    if (k != 0xffffu) {
      for (uint32_t i = k; i < jb.in_len; i += jb.out_len) {
        acc += x[i];
      }
    }
    return acc;
  };
  const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(inv)) & 15u) == 0;
  if (aligned && j0 + V <= out_span) {
    uint16_t k[V];
    if (V == 8) {
      const uint4 q = *reinterpret_cast<const uint4*>(inv + j0);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        k[i] = (uint16_t)(w[i >> 1] >> (16 * (i & 1)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < V; i++) {
        k[i] = inv[j0 + i];
      }
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < V; i++) {
      any = any || (k[i] != 0xffffu && k[i] < jb.in_len);
    }
    if (!any) {
      return;
    }
    uint4 cur = *reinterpret_cast<uint4*>(dst + j0);
    T*    e   = reinterpret_cast<T*>(&cur);
#pragma unroll
    for (int i = 0; i < V; i++) {
      e[i] = (T)(e[i] + gather(k[i]));
    }
    *reinterpret_cast<uint4*>(dst + j0) = cur;
  } else {
    for (uint32_t j = j0; j < j0 + V && j < out_span; j++) {
      const uint32_t k = inv[j];
      if (k != 0xffffu && k < jb.in_len) {
        dst[j] = (T)(dst[j] + gather(k));
      }
    }
  }
}

// Same operation with the code block's input staged in LDS: one workgroup per code block loads the E soft bits with
// coalesced 16-byte loads, then walks the soft buffer (16 bytes per lane, one table dwordx4, gathers from LDS, one
// read-modify-write).  The global gathers of the kernel above (2 bytes out of every line they touch) were its limit.
template <typename T>
__global__ __launch_bounds__(512) void rm_rx_gather_lds_kernel(const T* in, T* out, const uint16_t* inverse, const RxJob* jobs, const RxJob uni,
                                                               uint32_t in_stride, uint32_t out_stride, uint32_t out_span)
{
  extern __shared__ uint4 stage[];
  RxJob jb;
  if (jobs) {
    jb = jobs[blockIdx.x];
  } else {
    jb = uni;
    jb.in_offset += blockIdx.x * in_stride;
    jb.out_offset += blockIdx.x * out_stride;
  }
  constexpr int   V   = 16 / sizeof(T);
  const T*        x   = in + jb.in_offset;
  T*              dst = out + jb.out_offset;
  const uint16_t* inv = inverse + jb.table;
  T*              xs  = reinterpret_cast<T*>(stage);
  // stage: aligned middle part as uint4, ragged ends element-wise
  const uint32_t head = min(jb.in_len, (uint32_t)(((16u - ((uintptr_t)x & 15u)) & 15u) / sizeof(T)));
  const uint32_t nq   = (jb.in_len - head) / V;
  for (uint32_t i = threadIdx.x; i < head; i += blockDim.x) {
    xs[i] = x[i];
  }
  for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + head + q * V);
    T           t[V];
    *reinterpret_cast<uint4*>(t) = v;
#pragma unroll
    for (int i = 0; i < V; i++) {
      xs[head + q * V + i] = t[i]; // head shifts the alignment: element-wise LDS stores
    }
  }
  for (uint32_t i = head + nq * V + threadIdx.x; i < jb.in_len; i += blockDim.x) {
    xs[i] = x[i];
  }
  __syncthreads();
  auto gather = [&](uint32_t k) {
    int acc = 0;
    if (k != 0xffffu) {
      for (uint32_t i = k; i < jb.in_len; i += jb.out_len) {
        acc += xs[i];
      }
    }
    return acc;
  };
  const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(inv)) & 15u) == 0;
  for (uint32_t j0 = threadIdx.x * V; j0 < out_span; j0 += blockDim.x * V) {
    if (aligned && j0 + V <= out_span) {
      uint16_t k[V];
      if (V == 8) {
        const uint4    q    = *reinterpret_cast<const uint4*>(inv + j0);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
          k[i] = (uint16_t)(w[i >> 1] >> (16 * (i & 1)));
        }
      } else {
#pragma unroll
        for (int i = 0; i < V; i++) {
          k[i] = inv[j0 + i];
        }
      }
      bool any = false;
#pragma unroll
      for (int i = 0; i < V; i++) {
        any = any || (k[i] != 0xffffu && k[i] < jb.in_len);
      }
      if (!any && !jb.fresh) {
        continue;
      }
      uint4 cur = jb.fresh ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<uint4*>(dst + j0);
      T*    e   = reinterpret_cast<T*>(&cur);
#pragma unroll
      for (int i = 0; i < V; i++) {
        e[i] = (T)(e[i] + gather(k[i]));
      }
      *reinterpret_cast<uint4*>(dst + j0) = cur;
    } else {
      for (uint32_t j = j0; j < j0 + V && j < out_span; j++) {
        const uint32_t k = inv[j];
        if (k != 0xffffu && k < jb.in_len) {
          dst[j] = (T)((jb.fresh ? (T)0 : dst[j]) + gather(k));
        } else if (jb.fresh) {
          dst[j] = (T)0;
        }
      }
    }
  }
}

hipError_t launch_rx_gather(const void* d_in, void* d_out, const uint16_t* d_inverse, uint32_t out_span, const RxJob* d_jobs,
                            const RxJob& uni, uint32_t in_stride, uint32_t out_stride, int n_jobs, bool elem8, hipStream_t stream,
                            uint32_t max_in_len)
{
  const size_t stage_bytes = (((size_t)max_in_len * (elem8 ? 1 : 2)) + 15) & ~(size_t)15;
  if (max_in_len && stage_bytes <= 40 * 1024) { // up to 4 workgroups per CU
    dim3 grid(n_jobs);
    if (elem8) {
      hipLaunchKernelGGL(rm_rx_gather_lds_kernel<signed char>, grid, dim3(512), stage_bytes, stream, (const signed char*)d_in, (signed char*)d_out,
                         d_inverse, d_jobs, uni, in_stride, out_stride, out_span);
    } else {
      hipLaunchKernelGGL(rm_rx_gather_lds_kernel<short>, grid, dim3(512), stage_bytes, stream, (const short*)d_in, (short*)d_out, d_inverse, d_jobs,
                         uni, in_stride, out_stride, out_span);
    }
    return hipGetLastError();
  }
  const uint32_t per_wg = 256 * (elem8 ? 16 : 8);
  dim3           grid((out_span + per_wg - 1) / per_wg, n_jobs);
  if (elem8) {
    hipLaunchKernelGGL(rm_rx_gather_kernel<signed char>, grid, dim3(256), 0, stream, (const signed char*)d_in, (signed char*)d_out,
                       d_inverse, d_jobs, uni, in_stride, out_stride, out_span);
  } else {
    hipLaunchKernelGGL(rm_rx_gather_kernel<short>, grid, dim3(256), 0, stream, (const short*)d_in, (short*)d_out, d_inverse, d_jobs, uni,
                       in_stride, out_stride, out_span);
  }
  return hipGetLastError();
}

static hipError_t launch_any(const void* d_in, void* d_out, const uint16_t* d_tables, const RxJob* d_jobs, const RxJob& uni,
                             uint32_t in_stride, uint32_t out_stride, int n_jobs, bool elem8, hipStream_t stream)
{
  // out_len <= 3 * 6144 + 12 = 18444 -> 73 workgroups of 256 per code block
  dim3 grid(73, n_jobs);
  if (elem8) {
    hipLaunchKernelGGL(rm_rx_kernel<signed char>, grid, dim3(256), 0, stream, (const signed char*)d_in, (signed char*)d_out, d_tables,
                       d_jobs, uni, in_stride, out_stride);
  } else {
    hipLaunchKernelGGL(rm_rx_kernel<short>, grid, dim3(256), 0, stream, (const short*)d_in, (short*)d_out, d_tables, d_jobs, uni,
                       in_stride, out_stride);
  }
  return hipGetLastError();
}

hipError_t launch_rx(const void* d_in, void* d_out, const uint16_t* d_tables, const RxJob* d_jobs, int n_jobs, bool elem8,
                     hipStream_t stream)
{
  return launch_any(d_in, d_out, d_tables, d_jobs, RxJob{}, 0, 0, n_jobs, elem8, stream);
}

hipError_t launch_rx_uniform(const void* d_in, void* d_out, const uint16_t* d_table, const RxJob& first, uint32_t in_stride,
                             uint32_t out_stride, int n_jobs, bool elem8, hipStream_t stream)
{
  return launch_any(d_in, d_out, d_table, nullptr, first, in_stride, out_stride, n_jobs, elem8, stream);
}

} // namespace rm
} // namespace phyhip

namespace phyhip {
namespace rm {

// ---- transport-block CRC (decode_tb, sch.c:540-560): CRC24A over the tbs payload bits, compared with the three
// parity bytes that follow.  One workgroup per block: every lane runs the CRC (crc.c:92-140, a byte per table look-up) of its chunk,
// shifts it to its place by x^(8 * bytes behind the chunk) mod g (square-and-multiply) and the partial checksums
// are XOR-ed.
__device__ __forceinline__ uint32_t gf_mulmod(uint32_t a, uint32_t b, uint32_t poly)
{
  uint32_t r = 0;
  for (int i = 23; i >= 0; i--) {
    r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? poly : 0u);
    r ^= ((b >> i) & 1u) ? a : 0u;
  }
  return r;
}

// Only blocks whose code blocks all passed get a parity (sch.c:473-477), decided HERE from the decoder's verdicts: the host does not have
// to come back between decoding and this.  Every lane runs the CRC of its chunk a byte per table look-up (256-entry table of the generator,
// built in LDS), multiplies it into place with the host's x^(8 * bytes behind the chunk) mod g for this block size, and the partial
// checksums are XOR-ed.  (The square-and-multiply each lane used to do for that power was 3/4 of the kernel: 0.143 -> 0.03 ms per 2944 blocks.)
__host__ __device__ static inline uint32_t tb_crc_chunk(uint32_t nb) { return (nb + 255) / 256; }

__global__ __launch_bounds__(256) void tb_crc_kernel(const uint8_t* data, const TbCrcJob* jobs, uint32_t poly_full, const uint8_t* cb_ok,
                                                     const uint32_t* mult, TbCrcResult* res, uint8_t* host_data)
{
  __shared__ uint32_t red[256];
  __shared__ uint32_t tab[256];
  __shared__ uint32_t s_good;
  const TbCrcJob jb   = jobs[blockIdx.x];
  const uint32_t poly = poly_full & 0xffffffu;
  if (host_data) {
    // single transport blocks on host buffers (decode_tb_cb): the decoded bytes -- payload, transport CRC and the last block's own CRC behind it -- go
    // to the caller's pinned image from here instead of a copy operation of their own behind this kernel (a launch more in a chain of seven)
    // (16 bytes per lane: a block's place in both images starts on a 256-byte boundary and is padded to one, sch_host.cpp)
    const uint32_t nvec = (jb.tbs / 8 + 6 + 15) / 16;
    const uint4*   src  = reinterpret_cast<const uint4*>(data + jb.data_offset);
    uint4*         dst  = reinterpret_cast<uint4*>(host_data + jb.data_offset);
    for (uint32_t i = threadIdx.x; i < nvec; i += 256) {
      dst[i] = src[i];
    }
  }
  {
    uint32_t c = (uint32_t)threadIdx.x << 16;
    for (int b = 0; b < 8; b++) {
      c = ((c << 1) & 0xffffffu) ^ ((c & 0x800000u) ? poly : 0u);
    }
    tab[threadIdx.x] = c;
  }
  if (threadIdx.x == 0) {
    s_good = 0;
  }
  __syncthreads();
  {
    uint32_t good = 0;
    for (int k = 0; k < 2; k++) {
      for (uint32_t i = threadIdx.x; i < jb.run_len[k]; i += 256) {
        good += cb_ok[jb.run_start[k] + i] ? 1u : 0u;
      }
    }
    if (good) {
      atomicAdd(&s_good, good);
    }
  }
  __syncthreads();
  if (s_good != jb.need) {
    if (threadIdx.x == 0) {
      res[blockIdx.x] = {0u, 0u, 0u};
    }
    return;
  }
  const uint8_t* d    = data + jb.data_offset;
  const uint32_t nb   = jb.tbs / 8;
  const uint32_t c    = tb_crc_chunk(nb);
  const uint32_t lo = threadIdx.x * c, hi = lo + c < nb ? lo + c : nb;
  uint32_t       crc = 0;
  for (uint32_t i = lo; i < hi; i++) {
    crc = ((crc << 8) & 0xffffffu) ^ tab[((crc >> 16) ^ d[i]) & 0xffu];
  }
  red[threadIdx.x] = lo < nb ? gf_mulmod(crc, mult[256u * jb.mult + threadIdx.x], poly) : 0u;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x] ^= red[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t par_rx = red[0];
    const uint32_t par_tx = ((uint32_t)d[nb] << 16) | ((uint32_t)d[nb + 1] << 8) | (uint32_t)d[nb + 2];
    res[blockIdx.x]       = {par_rx, par_tx, 1u};
  }
}

hipError_t launch_tb_crc(const uint8_t* d_data, const TbCrcJob* d_jobs, int n_jobs, uint32_t poly, const uint8_t* d_cb_ok, const uint32_t* d_mult,
                         TbCrcResult* d_res, hipStream_t stream, uint8_t* host_data)
{
  hipLaunchKernelGGL(tb_crc_kernel, dim3(n_jobs), dim3(256), 0, stream, d_data, d_jobs, poly, d_cb_ok, d_mult, d_res, host_data);
  return hipGetLastError();
}

void tb_crc_multipliers(uint32_t tbs, uint32_t poly_full, uint32_t out[256])
{
  const uint32_t poly = poly_full & 0xffffffu;
  auto           mul  = [poly](uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int i = 23; i >= 0; i--) {
      r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? poly : 0u);
      r ^= ((b >> i) & 1u) ? a : 0u;
    }
    return r;
  };
  auto xpow = [&](uint32_t e) { // x^e mod g
    uint32_t result = 1, base = 2;
    while (e) {
      if (e & 1) {
        result = mul(result, base);
      }
      base = mul(base, base);
      e >>= 1;
    }
    return result;
  };
  const uint32_t nb = tbs / 8, c = tb_crc_chunk(nb);
  const uint32_t step = xpow(8 * c);
  uint32_t       m = 1, behind = 0; // from the last lane backwards: bytes behind lane l = bytes of the lanes after it
  for (int l = 255; l >= 0; l--) {
    const uint32_t lo = (uint32_t)l * c, hi = lo + c < nb ? lo + c : nb;
    if (lo >= nb) {
      out[l] = 0;
      continue;
    }
    // behind = nb - hi
    if (nb - hi != behind) {
      m      = (nb - hi == behind + c) ? mul(m, step) : xpow(8 * (nb - hi));
      behind = nb - hi;
    }
    out[l] = m;
  }
}

} // namespace rm
} // namespace phyhip
