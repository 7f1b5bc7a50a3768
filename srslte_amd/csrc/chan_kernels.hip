// chan_kernels.hip -- see chan_device.h (gfx950)
#include "chan_device.h"
#include "hip_common.h"

namespace phyhip {
namespace chan {

namespace {
// one lane per OUTPUT byte: its 8 bits come from up to 8 different groups of g (a gather of single bits from an array that fits the caches:
// a grant has at most 86,400 bits); the writes are coalesced and every byte is written exactly once, so q needs no clearing
__global__ __launch_bounds__(256) void ul_interleave_bits_kernel(const uint8_t* __restrict__ g, uint8_t* __restrict__ q, uint32_t nof_sym, uint32_t Qm,
                                                                 uint32_t rows, uint32_t cols)
{
  const uint32_t nbits = nof_sym * Qm;
  const uint32_t byte  = blockIdx.x * 256u + threadIdx.x;
  if (byte * 8u >= nbits) {
    return;
  }
  uint32_t v = 0;
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) {
    const uint32_t b = byte * 8u + i;
    if (b < nbits) {
      const uint32_t s = b / Qm, k = b - s * Qm; // group of q and bit inside it
      const uint32_t c = s / rows, r = s - c * rows;
      const uint32_t src = (r * cols + c) * Qm + k;
      v |= ((g[src >> 3] >> (7u - (src & 7u))) & 1u) << (7u - i);
    }
  }
  q[byte] = (uint8_t)v;
}
} // namespace

hipError_t launch_ul_interleave_bits(const uint8_t* g_bits, uint8_t* q_bits, uint32_t nof_sym, uint32_t Qm, uint32_t cols, hipStream_t stream)
{
  if (nof_sym == 0 || Qm == 0 || cols == 0 || nof_sym % cols) {
    return hipErrorInvalidValue;
  }
  const uint32_t nbytes = (nof_sym * Qm + 7) / 8;
  hipLaunchKernelGGL(ul_interleave_bits_kernel, dim3(ceil_div(nbytes, 256u)), dim3(256), 0, stream, g_bits, q_bits, nof_sym, Qm, nof_sym / cols, cols);
  return hipGetLastError();
}

} // namespace chan
} // namespace phyhip
