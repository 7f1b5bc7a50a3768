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
