// turbo_device.h -- kernel parameter blocks and launchers of turbo_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace turbo {

// optional per-code-block placement (transport-block decoding): where the LLRs of a block start, where its hard bits
// go and how many of its K/8 bytes are written (sch.c:424: all but the last block of a transport block drop their CRC)
struct CbDesc {
  uint32_t in_off;    // elements from WinParams::input
  uint32_t out_off;   // bytes from WinParams::output
  uint32_t out_bytes; // <= K/8
  uint32_t reserved;
};

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
  // early stop on CRC (decode_tb_cb, sch.c:420-454): generator incl. the x^24 term, 0 = off
  const CbDesc*   desc;     // optional, n_cb entries
  uint32_t        crc_poly;
  const uint32_t* crc_mult; // nb multipliers x^(W (nb-1-d)) mod g
  int*            noi;      // out: half iterations run per code block
  uint8_t*        crc_ok;   // out: 1 = CRC matched
  // measured launch-shape alternatives (SRSRAN_HIP_TDEC_VARIANT, never the default): 0 product, 1 one wave per SIMD, 2 persistent grid
  int             variant;
  uint32_t        n_units;      // persistent grid: units of 64 / (nb / 2) code blocks,
  uint32_t        max_resident; //   workgroups launched at most,
  uint32_t*       unit_counter; //   device counter, zero at launch
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
  // transport-block decoding (as WinParams): per-block placement and CRC early stop, 0 / null = off
  const CbDesc*   desc;
  uint32_t        crc_poly;
  int*            noi;
  uint8_t*        crc_ok;
  const uint32_t* crc_mult8; // latency kernel with crc_poly: x^(bits behind the eighth of lane l) mod g, 8 words (turbo_host.cpp: gen_crc_mult8)
};
// latency kernel of the scalar decoder (turbo_gen_lat_kernels.hip): 8 lanes per code block, the block's state in LDS; same GenParams, the workspace
// (what a resumed run reads back) laid out per block: 15 (K + 4) int16 apart, seven arrays of K + 4
hipError_t launch_gen_lat(const GenParams& p, hipStream_t stream);
size_t     gen_lat_lds_bytes(uint32_t K);
constexpr uint32_t kGenLatMaxBlocks = 8192; // (K = 400: 8192 blocks 0.33 against 0.51 ms per half iteration, 32768 blocks 1.36 against 0.69: tools/measure/gen_time.py)

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
// latency kernel (turbo_lat_kernels.hip): one code block per wave, states across lanes; same WinParams with its own workspace layout
// (lat_ws_dwords per code block).  Exists for every window decoder: 16 / 8 sub-blocks (16-bit), 32 / 16 sub-blocks (8-bit; 32 sub-blocks = two waves per block).
hipError_t launch_lat(int nb, bool arith8, const WinParams& p, hipStream_t stream);
// its two-wave form (16 sub-blocks: K > 800 with 16-bit soft bits, 800 < K <= 2048 with 8-bit ones): forward and backward recursion of a block at the same time, one wave each; same workspace layout,
// so a run may change between the two forms from launch to launch
hipError_t launch_lat2(bool arith8, const WinParams& p, hipStream_t stream);
size_t     lat2_lds_bytes(uint32_t K);
constexpr uint32_t kLat2MaxBlocks = 256; // per block-per-CU the filed rows allow (turbo_host.cpp: want_lat2)
uint32_t   lat_ws_dwords(uint32_t K, int nb);
static inline bool lat_exists(int nb, bool arith8)
{
  return (nb == 16) || (nb == 8 && !arith8) || (nb == 32 && arith8);
}
// Batches of up to this many WAVES go to the latency kernel (one wave per code block; two for the 32-sub-block 8-bit decoder): one wave per
// SIMD fills the chip's 1024 SIMDs once -- a lone wave already issues at the rate a SIMD sustains for packed / three-operand instructions
// (tools/probe/valu_issue_probe.hip) --, two waves per SIMD take twice as long each; beyond that the throughput kernel's 8 blocks per wave
// win.  Measured (K = 6144, 8 half iterations, profiles/r03_lat_time.txt): 16-bit 1024 blocks 0.62 against 1.81 ms, 2048 blocks 1.20 against
// 1.84, 4096 blocks 2.44 against 2.03; 8-bit (two waves per block) 512 blocks 0.47 against 1.18, 1024 blocks 1.00 against 1.21, 2048 blocks 1.99 against 1.32.
constexpr uint32_t kLatMaxBlocks = 2048; // in waves
static inline uint32_t lat_waves(int nb, uint32_t n_cb)
{
  return nb == 32 ? 2 * n_cb : n_cb;
}
} // namespace turbo
} // namespace phyhip
struct srsran_hip_tdec_batch;
namespace phyhip {
namespace turbo {
// One workspace for all the decoders of a transport-block stage (sch_host.cpp): its launches run one after the other on the stage's stream and every
// launch is a complete run, so the decoders of different block sizes can work in the same memory -- a block size the stage has not seen yet then
// needs no allocation.  ensure() grows the arena (after waiting for `stream`, whose launches may still be using the old one).
struct WsArena {
  void*  p   = nullptr;
  size_t cap = 0;
  void*  ensure(size_t bytes, hipStream_t stream);
  ~WsArena();
};
// decoder object (AUTO implementation) whose workspace is the arena's
int batch_create_shared(srsran_hip_tdec_batch** h, uint32_t long_cb, uint32_t max_nof_cb, bool llr8_api, WsArena* arena);
// every table a decoder of any block size needs (exchange / interleaver tables, CRC multipliers of both generators), in one allocation
bool prebuild_tables();
// host side (turbo_host.cpp), used by the transport-block decoder
int batch_run_early_stop(srsran_hip_tdec_batch* h, const void* d_input, bool in_is8, const CbDesc* d_desc, uint8_t* d_output,
                         uint32_t n_cb, uint32_t max_iterations, int sb_layout, uint32_t crc_poly, int* d_noi, uint8_t* d_crc_ok,
                         hipStream_t stream);
hipError_t launch_gen(const GenParams& p, hipStream_t stream);
uint32_t   win_elem_index(int nb, uint32_t k, uint32_t d);

} // namespace turbo
} // namespace phyhip
