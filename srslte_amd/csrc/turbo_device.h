// turbo_device.h -- kernel parameter blocks and launchers of turbo_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace turbo {

struct WinParams {
  const short*    input;   // n_cb code blocks, in_stride elements apart (int8 elements when in_is8)
  uint8_t*        output;  // n_cb x K/8 bytes, out_stride apart
  short*          dec_llr; // optional: n_cb x K decision LLRs, natural order
  uint32_t*       ws;      // per-code-block workspace, ws_stride dwords apart
  const uint32_t* deint;   // blocked-layout scatter table for app2[deinter[i]] = ext1[i]
  const uint32_t* inter;   // blocked-layout scatter table for app1[inter[i]]  = ext2[i]
  uint32_t        in_stride;
  uint32_t        out_stride;
  uint32_t        ws_stride;
  uint32_t        K;
  uint32_t        n_begin; // first half iteration of this launch (0: also extract the input)
  uint32_t        n_end;   // one past the last half iteration; the hard decision is taken for n_iter = n_end
  int             n_cb;
  int             sb_layout;
  int             in_is8;
};

struct GenParams {
  const short*    input;
  uint8_t*        output;
  short*          dec_llr;
  short*          ws; // per-wave slab, ws_stride int16 apart
  const uint16_t* inter;
  const uint16_t* deinter;
  size_t          ws_stride;
  uint32_t        in_stride;
  uint32_t        out_stride;
  uint32_t        K;
  uint32_t        n_begin;
  uint32_t        n_end;
  int             n_cb;
  int             in_is8;
};

// dwords of workspace per code block for the window decoder with nb sub-blocks
static inline uint32_t win_ws_dwords(uint32_t K, int nb)
{
  uint32_t lpc = nb / 2, long_sb = K / nb, nblk = (long_sb + 7) / 8;
  return 6 * nblk * lpc * 8 + (nblk + 1) * lpc * 8 + 8 * lpc; // (+ tail LLRs); per code block, the kernel interleaves a wave's blocks
}
// int16 of workspace per wave (64 code blocks) for the scalar decoder
static inline size_t gen_ws_shorts(uint32_t K)
{
  return (size_t)15 * (K + 4) * 64;
}

hipError_t launch_win(int nb, bool arith8, const WinParams& p, hipStream_t stream);
hipError_t launch_gen(const GenParams& p, hipStream_t stream);
uint32_t   win_elem_index(int nb, uint32_t k, uint32_t d);

} // namespace turbo
} // namespace phyhip
