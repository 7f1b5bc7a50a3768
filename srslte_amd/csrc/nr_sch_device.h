// nr_sch_device.h -- parameter blocks and launchers of nr_sch_kernels.hip: NR LDPC rate matching (both directions) and
// the LDPC encoder (lib/src/phy/fec/ldpc/ldpc_rm.c, ldpc_encoder.c, ldpc_enc_c.c)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace nrsch {

enum { T_I8 = 0, T_I16 = 1, T_F32 = 2 };

struct CbJob { // one code block
  uint32_t in_off;  // elements
  uint32_t out_off; // elements
  uint32_t E;       // rate-matched length
  uint32_t aux;     // encoder: cdwd_rm_length after clamping / rounding; de-matcher: bit 0 = new data (overwrite, do not accumulate)
};

struct RmParams { // init_rm, ldpc_rm.c:113-167
  const void*  in;
  void*        out;
  const CbJob* jobs;
  uint32_t     n_cb;
  uint32_t     Ncb, k0, ini_ex, end_ex; // circular buffer, start, filler positions [ini_ex, end_ex)
  uint32_t     Qm;
  int          type;
};

// max_E: the largest job.E of the batch; min_E_new: the smallest job.E among the jobs flagged as new data (aux bit 0), ~0u if none
hipError_t launch_rm_rx(const RmParams& p, uint32_t max_E, hipStream_t stream, uint32_t min_E_new = ~0u);
hipError_t launch_rm_tx(const RmParams& p, hipStream_t stream); // uint8 code words -> uint8 rate-matched bits

#define NRSCH_MAX_CORE_TERMS 4
struct EncStep { // one of the three core parity blocks solved after p0: p[unk] = rot(lam[row] + sum rot(p[blk], sh), -ush)
  int row, unk, ush;
  int n_terms;
  int blk[NRSCH_MAX_CORE_TERMS], sh[NRSCH_MAX_CORE_TERMS];
};

struct EncParams {
  const uint8_t* in;  // n_cb messages: bgK * Z bytes (bit per byte, 254 = filler)
  uint8_t*       out; // n_cb code words: (bgN - 2) * Z bytes
  const CbJob*   jobs;
  uint32_t       n_cb;
  const int*     row_start; // bgM + 1
  const int*     edges;     // col | shift << 8
  int            Z, bgN, bgM, bgK;
  int            a;  // p0 = rot(lam0 + lam1 + lam2 + lam3, -a)
  EncStep        step[3];
};

hipError_t launch_encode(const EncParams& p, hipStream_t stream);

// ---- transmit side of the transport-block loop (sch_nr_encode, sch_nr.c:375-520), in front of the encoder
struct TbEnc {
  uint32_t payload_off, tbs, L_tb;
};
struct CbEnc {
  uint32_t tb;       // index into the TbEnc array
  uint32_t bit_off;  // first payload bit of this code block
  uint32_t cb_len;   // payload bits it carries (:431-446)
  uint32_t Kp, Kr, L_cb, last;
  uint32_t msg_row;
};
// transport CRC of every payload (:426), then the messages: payload bits, transport CRC behind the last block, CRC24B, filler marks
hipError_t launch_tb_crc_enc(const uint8_t* d_payload, const TbEnc* d_tbs, uint32_t n_tb, uint32_t* d_crc, hipStream_t stream);
hipError_t launch_cb_build(const uint8_t* d_payload, const CbEnc* d_cbs, uint32_t n_cb, const TbEnc* d_tbs, const uint32_t* d_crc, uint8_t* d_msg,
                           uint32_t msg_stride, hipStream_t stream);

// ---- transport-block loop of sch_nr_decode (sch_nr.c:620-712) behind the decoder
struct CbFin {       // one code block that has just been through the decoder
  uint32_t msg_row;  // row of the decoder's message array (one bit per byte)
  uint32_t cb_index; // global code-block index: flag and packed-data row
  uint32_t cb_len;   // Kp - L_cb bits (a multiple of 8)
};
struct TbFin {
  uint32_t first_cb, C, Kp, L_cb, L_tb, tbs, payload_off;
  uint32_t mult; // row of the CRC multiplier table (tb_finish_multipliers) for this block's chunk size and CRC
};
struct TbFinRes {
  int32_t all_decoded; // every code block of the transport block has its flag set
  int32_t crc_ok;      // :696-705 (single code block: true; otherwise the transport CRC over the payload matches)
};
// flag = (iterations != 0) && !all_zeros (:633-639); flagged blocks are packed MSB first into their data row (:650-652)
hipError_t launch_cb_finish(const uint8_t* d_msg, uint32_t msg_stride, const CbFin* d_jobs, const int* d_n_iter, uint32_t n, uint8_t* d_flags,
                            uint8_t* d_cb_data, uint32_t data_stride, hipStream_t stream);
// d_mult: rows of 256 lane multipliers, one per (chunk size, CRC order) of the launch
hipError_t launch_tb_finish(const uint8_t* d_cb_data, uint32_t data_stride, const uint8_t* d_flags, const TbFin* d_jobs, uint32_t n, uint8_t* d_payload,
                            const uint32_t* d_mult, TbFinRes* d_res, hipStream_t stream);
// The kernel gives each of its 256 lanes `chunk` = ceil(bytes / 256) bytes of the payload PADDED IN FRONT with zeros (they do not change a CRC that
// starts at zero), so that lane l always has (255 - l) * chunk bytes behind it: out[l] = x^(8 chunk (255 - l)) mod g.
uint32_t tb_finish_chunk(uint32_t tbs_bits);
void     tb_finish_multipliers(uint32_t chunk, uint32_t order, uint32_t out[256]);

} // namespace nrsch
} // namespace phyhip
