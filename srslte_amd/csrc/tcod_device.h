// tcod_device.h -- parameter blocks and launchers of tcod_kernels.hip: LTE turbo encoder and the transmit side of a transport
// block (CRC attachment, segmentation, turbo coding, rate matching, concatenation; sch.c encode_tb / turbocoder.c / rm_turbo.c)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace tcod {

struct EncParams { // srsran_tcod_encode over a batch of code blocks of one size
  const uint8_t* in;  // n_cb x K bytes (bit per byte; 100 = SRSRAN_TX_NULL), in_stride apart
  uint8_t*       out; // n_cb x (3K + 12) bytes, out_stride apart
  uint32_t       in_stride, out_stride;
  uint32_t       n_cb, K, f1, f2;
};
hipError_t launch_encode(const EncParams& p, hipStream_t stream);

struct TbCbJob { // one code block of a transport block
  uint32_t src_bit;    // first payload bit of this block in d_data (bit offset; payload bytes are MSB first)
  uint32_t n_src_bits; // payload bits taken from d_data
  uint32_t tb_crc;     // index into tb_crc[] when the 24 transport-block CRC bits follow the payload in this block, else 0xffffffff
  uint32_t crc24b;     // 1: append CRC24B (more than one block in the transport block)
  uint32_t K, f1, f2;  // block size (= n_src_bits + 24 per CRC) and its QPP parameters
  uint32_t E;          // rate-matched bits
  uint32_t out_bit;    // first output bit in d_e (bit offset, MSB first)
  const uint16_t* table; // forward rate-matching table of (K, rv): position in the natural [d0 d1 d2] x K + 12 buffer of each bit
  uint32_t table_len;
  uint32_t pad;
  uint32_t crc_mult_row; // CRC24B: row of TbParams::crc_mult with lane l's x^(bits behind its stretch) mod g for a block of this many bits
  uint32_t crc_mult256_row; // latency kernel: row of TbParams::crc_mult256 (256 lanes, stretches counted from the END of the message)
};
struct TbCrcJob { // CRC24A of one transport block
  uint32_t src_byte, n_bytes;
  uint32_t crc_mult_row; // row of TbParams::crc_mult with lane l's x^(8 bytes behind its stretch) mod g for this many bytes
  uint32_t crc_mult256_row; // latency kernel: row of TbParams::crc_mult256
};
// the lanes' stretches of an n-unit message and what lies behind them: lane l takes units [min(l L, n), min((l + 1) L, n)), L = ceil(n / 64)
void crc_lane_multipliers(uint32_t n_units, uint32_t bits_per_unit, uint32_t poly24, uint32_t* m64);
struct TbParams {
  const uint8_t*  data;   // packed payload bytes of all transport blocks
  uint8_t*        e_bits; // packed output, zeroed by the launcher's caller (blocks OR their bits in)
  const TbCbJob*  cbs;
  const TbCrcJob* tbs;
  uint32_t*       tb_crc; // n_tb checksums (device scratch)
  const uint32_t* crc_mult; // rows of 64 lane multipliers (host: crc_lane_multipliers)
  uint32_t        n_cb, n_tb;
  const uint32_t* crc_mult256; // latency kernel: rows of 256 lane multipliers (host: crc_lane_multipliers256)
};
// Latency kernel's split of an n-unit message over 256 lanes: stretches of L = ceil(n / 256) units counted from the END (lane 255 takes the last L
// units, lane l the units [n - (256 - l) L, n - (255 - l) L) clipped at 0 -- leading zeros do not change a CRC with a zero start), so every lane has
// (255 - l) L units behind it: m[l] = x^(bits_per_unit L (255 - l)) mod g.
void crc_lane_multipliers256(uint32_t n_units, uint32_t bits_per_unit, uint32_t poly24, uint32_t* m256);
// One launch for a transport block (or a subframe's worth of them): one workgroup of 256 lanes per code block -- payload bits, transport CRC24A (by the
// workgroup of the block that carries it), CRC24B, both constituent encoders, rate matching.  Same TbParams; tb_crc scratch unused.
hipError_t launch_tb_encode_lat(const TbParams& p, hipStream_t stream);
hipError_t launch_tb_crc24a(const TbParams& p, hipStream_t stream);

struct LutEncParams { // srsran_tcod_encode_lut for one code block
  const uint8_t* in;      // n_data_bytes packed payload bytes
  uint8_t*       out_sys; // K / 8 + 1 bytes: the block with its CRCs, then the systematic tail nibble
  uint8_t*       out_par; // K / 4 + 1 bytes: [parity 0 | tail | parity 1 | tail]
  uint32_t*      crc_state; // [0]: transport-block checksum in / out; [1]: code-block checksum out
  uint32_t       K, f1, f2, n_data_bytes;
  uint32_t       tb_poly, cb_poly;
  uint32_t       has_cb_crc, last_cb;
};
hipError_t launch_lut_encode(const LutEncParams& p, hipStream_t stream);

struct LutRmParams { // srsran_rm_turbo_tx_lut for one code block
  const uint8_t*  sys;
  const uint8_t*  par;
  uint8_t*        out; // E bits packed from bit 0
  const uint16_t* table;
  uint32_t        table_len, K, E;
};
hipError_t launch_lut_rm(const LutRmParams& p, hipStream_t stream);
hipError_t launch_tb_encode(const TbParams& p, hipStream_t stream);

} // namespace tcod
} // namespace phyhip
