/*
 * phy_sch_abi.h -- the step in front of the turbo decoder: rate de-matching.
 *
 * Drop-in entry points of lib/include/srsran/phy/fec/turbo/rm_turbo.h:76-89 (receive side) plus a device-resident
 * batched form.  The output layout follows the reference: srsran_rm_turbo_rx_lut writes the sub-block layout the
 * window decoder of srsran_tdec_autoimp_get_subblocks(K) reads directly (rm_turbo.c:248-273,405-421), or the natural
 * buffer [d0 d1 d2] x (K+4) when that decoder is the scalar one.
 */
#ifndef SRSRAN_AMD_PHY_SCH_ABI_H
#define SRSRAN_AMD_PHY_SCH_ABI_H

#include "srsran_amd/phy_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rm_turbo.h:76-89.  output (3(K+32)+12 int16 / int8) is accumulated into: HARQ combining. */
SRSRAN_API int  srsran_rm_turbo_rx_lut(int16_t* input, int16_t* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv_idx);
SRSRAN_API int  srsran_rm_turbo_rx_lut_(int16_t* input, int16_t* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv_idx,
                                        bool enable_input_tdec);
SRSRAN_API int  srsran_rm_turbo_rx_lut_8bit(int8_t* input, int8_t* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv_idx);
/* rm_turbo.h:49-51: the reference builds all 192 x 4 x 4 tables here; the HIP engine builds a table on first use */
SRSRAN_API void srsran_rm_turbo_gentables(void);
SRSRAN_API void srsran_rm_turbo_free_tables(void);

/* position of the k-th transmitted soft bit of redundancy version rv_idx in the receiver buffer (3K+12 entries, host
 * memory): the reference's static deinterleaver[cb][rv] (nof_sb = 0) / deinterleaver_sb[.][cb][rv] (8, 16, 32) tables */
SRSRAN_API int  srsran_hip_rm_turbo_table(uint16_t* table, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb);

/* ---- batched, device resident: n_cb code blocks of one size K and redundancy version, soft bits and soft buffers
 * `in_stride` / `out_stride` elements apart.  nof_sb: 0 natural buffer, 8 / 16 / 32 decoder sub-block layout. */
SRSRAN_API int srsran_hip_rm_turbo_rx_batch(const int16_t* d_input, uint32_t in_stride, uint32_t in_len, int16_t* d_softbuf,
                                            uint32_t out_stride, uint32_t n_cb, uint32_t long_cb, uint32_t rv_idx,
                                            uint32_t nof_sb, void* stream);
SRSRAN_API int srsran_hip_rm_turbo_rx_batch_8bit(const int8_t* d_input, uint32_t in_stride, uint32_t in_len, int8_t* d_softbuf,
                                                 uint32_t out_stride, uint32_t n_cb, uint32_t long_cb, uint32_t rv_idx,
                                                 uint32_t nof_sb, void* stream);

/* ---- transport-block decoding: decode_tb / decode_tb_cb of lib/src/phy/phch/sch.c:370-560 for a batch of blocks,
 * device resident.  Per code block: rate de-matching into its soft buffer (HARQ combining), turbo half iterations with
 * a CRC check after each one (CRC24B, or CRC24A when the block has a single code block) and early stop; then the
 * CRC24A of the whole block.  Differences from the reference's srsran_softbuffer_rx_t bookkeeping: the soft buffers, the
 * per-code-block CRC flags and the decoded bytes are arrays the CALLER keeps between HARQ rounds (a code block whose flag
 * is set is skipped and its bytes are expected to still be in d_data). */
#define SRSRAN_HIP_SOFTBUFFER_CB_SIZE 18600 /* int16 per code-block slot (SOFTBUFFER_SIZE, softbuffer.h) */

typedef struct {
  uint32_t tbs;         /* transport block size in bits; srsran_cbsegm must need no filler bits (sch.c:521) */
  uint32_t Qm;          /* bits per modulation symbol */
  uint32_t rv;          /* redundancy version 0..3 */
  uint32_t nof_e_bits;  /* soft bits of this block */
  uint32_t e_offset;    /* first soft bit in d_e_bits */
  uint32_t data_offset; /* first decoded byte in d_data; tbs/8 + 6 bytes are written */
  uint32_t first_cb;    /* first code-block slot (soft buffer, cb_crc) of this block */
} srsran_hip_tb_t;
/* OR-ed into srsran_hip_tb_t.rv: the block is new data (the caller would srsran_softbuffer_rx_reset it before decode_tb): its soft
 * buffer rows are overwritten by the rate de-matcher instead of accumulated into, so they need not be cleared beforehand; the
 * cb_crc flags of the block must be 0. */
#define SRSRAN_HIP_TB_NEW_DATA 0x100u

typedef struct {
  int32_t  crc_ok;         /* SRSRAN_SUCCESS or SRSRAN_ERROR: what decode_tb returns */
  float    avg_iterations; /* sch.c:485: half iterations per code block of the block */
  uint32_t nof_cb;
} srsran_hip_tb_result_t;

typedef struct srsran_hip_sch srsran_hip_sch_t;

SRSRAN_API int  srsran_hip_sch_create(srsran_hip_sch_t** h);
SRSRAN_API void srsran_hip_sch_free(srsran_hip_sch_t* h);
/* cb_crc: host array of flags per code-block slot, read and updated.  results: host array, n_tb entries.
 * Synchronises the stream before returning (the flags and results are host data). */
SRSRAN_API int  srsran_hip_sch_decode(srsran_hip_sch_t* h, const int16_t* d_e_bits, const srsran_hip_tb_t* tbs, uint32_t n_tb,
                                      uint32_t max_iterations, int16_t* d_softbuf, uint8_t* cb_crc, uint8_t* d_data,
                                      srsran_hip_tb_result_t* results, void* stream);
/* the same with q->llr_is_8bit (sch.c:408-412,426-428): int8 LLRs and soft buffers (same element counts and offsets), 8-bit rate
 * de-matching and the 8-bit window decoders */
SRSRAN_API int  srsran_hip_sch_decode_8bit(srsran_hip_sch_t* h, const int8_t* d_e_bits, const srsran_hip_tb_t* tbs, uint32_t n_tb,
                                           uint32_t max_iterations, int8_t* d_softbuf, uint8_t* cb_crc, uint8_t* d_data,
                                           srsran_hip_tb_result_t* results, void* stream);
/* srsran_cbsegm (cbsegm.h:32-45, cbsegm.c:62-117) */
typedef struct SRSRAN_API {
  uint32_t F, C, K1, K2, K1_idx, K2_idx, C1, C2, tbs, L_tb, L_cb, Z;
} srsran_cbsegm_t;

/* ---- the reference's OWN seam for a whole transport block: `bool decode_tb_cb(srsran_sch_t*, srsran_softbuffer_rx_t*, srsran_cbsegm_t*,
 * Qm, rv, nof_e_bits, void* e_bits, uint8_t* data)`, lib/src/phy/phch/sch.c:370-492 -- the only non-static function between
 * srsran_dlsch_decode2 / srsran_ulsch_decode (sch.c:579-618, 1121-1192) and the per-code-block calls.  Same arguments, same side
 * effects on HOST memory: e_bits are the caller's int16 (int8 when q->llr_is_8bit) soft bits, the soft buffer rows buffer_f[cb] receive
 * the combined, de-matched soft bits of every code block that was still undecoded (HARQ state for the next transmission), `data`
 * the decoded bytes (K/8 per undecoded code block at cb * rlen / 8, stored bytes of earlier rounds for the others), cb_crc / tb_crc /
 * data[cb] of the soft buffer and q->avg_iterations are updated as sch.c:448-452,472-486 do.  One call = one upload, one
 * de-matching launch and one early-stop decoder launch per block size, one download (the per-thread staging context is private).
 * Rows that arrive all zero (srsran_softbuffer_rx_reset, softbuffer.c:147-167) are not uploaded.
 * The two structs are the reference's layouts (softbuffer.h:40-47; sch.h:51-57 up to llr_is_8bit -- nothing behind it is touched). */
typedef struct SRSRAN_API {
  uint32_t  max_cb;
  uint32_t  max_cb_size;
  int16_t** buffer_f;
  uint8_t** data;
  bool*     cb_crc;
  bool      tb_crc;
} srsran_softbuffer_rx_t;
typedef struct SRSRAN_API {
  uint32_t max_iterations;
  float    avg_iterations;
  bool     llr_is_8bit;
} srsran_hip_sch_head_t;
SRSRAN_API bool srsran_hip_decode_tb_cb(void* q /* srsran_sch_t* */, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm,
                                        uint32_t rv, uint32_t nof_e_bits, void* e_bits, uint8_t* data);
/* the same under the reference's name, for a build whose sch.c no longer defines it */
SRSRAN_API bool decode_tb_cb(void* q, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv,
                             uint32_t nof_e_bits, void* e_bits, uint8_t* data);
SRSRAN_API int srsran_cbsegm(srsran_cbsegm_t* s, uint32_t tbs);
/* cbsegm.h:59-67, cbsegm.c:142-150,201-285: valid turbo block size; NR (LDPC) segmentation for base graph 1 / 2 */
SRSRAN_API bool srsran_cbsegm_cbsize_isvalid(uint32_t size);
SRSRAN_API int  srsran_cbsegm_ldpc_bg1(srsran_cbsegm_t* s, uint32_t tbs);
SRSRAN_API int  srsran_cbsegm_ldpc_bg2(srsran_cbsegm_t* s, uint32_t tbs);

/* ---- transmit side: turbo encoder (lib/include/srsran/phy/fec/turbo/turbocoder.h:46-58, turbocoder.c:40-185) ----
 * srsran_tcod_encode: input long_cb bits (one per byte; 100 = SRSRAN_TX_NULL filler, encoded as 0 and passed through on the
 * systematic and first parity outputs), output 3 long_cb + 12 bytes: [d0 d1 d2] per bit, then the 12 tail bits.
 * srsran_tcod_encode_lut / srsran_rm_turbo_tx_lut (turbocoder.h:60-66, rm_turbo.h:52-59) are the byte-packed per-block pair
 * encode_tb (sch.c:230-330) chains: same arguments, outputs and return values; the w_buff contents between redundancy versions
 * are private (as the reference's are).  The batched srsran_hip_sch_encode below is the throughput path. */
typedef struct SRSRAN_API {
  uint32_t max_long_cb;
  uint8_t* temp;
} srsran_tcod_t;
SRSRAN_API int  srsran_tcod_init(srsran_tcod_t* h, uint32_t max_long_cb);
SRSRAN_API void srsran_tcod_free(srsran_tcod_t* h);
SRSRAN_API int  srsran_tcod_encode(srsran_tcod_t* h, uint8_t* input, uint8_t* output, uint32_t long_cb);
SRSRAN_API void srsran_tcod_gentable(void);
/* input: the block's payload bytes; on return it also holds the CRC bytes this call appended and, in byte long_cb / 8, the
 * systematic tail nibble.  parity: long_cb / 4 + 1 bytes.  crc_tb: running CRC24A state; crc_cb: CRC24B object or NULL.
 * Returns 3 long_cb + 12, or -1. */
SRSRAN_API int  srsran_tcod_encode_lut(srsran_tcod_t* h, srsran_crc_t* crc_tb, srsran_crc_t* crc_cb, uint8_t* input, uint8_t* parity,
                                       uint32_t cblen_idx, bool last_cb);
SRSRAN_API int  srsran_rm_turbo_tx_lut(uint8_t* w_buff, uint8_t* systematic, uint8_t* parity, uint8_t* output, uint32_t cb_idx, uint32_t out_len,
                                       uint32_t w_offset, uint32_t rv_idx);
/* device resident: n_cb code blocks of long_cb bits, strides in bytes */
SRSRAN_API int srsran_hip_tcod_encode_batch(const uint8_t* d_in, uint32_t in_stride, uint8_t* d_out, uint32_t out_stride, uint32_t n_cb,
                                            uint32_t long_cb, void* stream);

/* encode_tb (sch.c:230-330) for a batch of transport blocks, device resident: CRC24A, segmentation, CRC24B per block, turbo
 * coding, rate matching of redundancy version rv, concatenation.  srsran_hip_tb_t: data_offset = first payload BYTE in d_data
 * (tbs / 8 bytes are read), e_offset = first output BIT in d_e_bits (MSB first; nof_e_bits bits are written, the range is
 * cleared first), first_cb unused.  Asynchronous on `stream`. */
typedef struct srsran_hip_sch_enc srsran_hip_sch_enc_t;
SRSRAN_API int  srsran_hip_sch_enc_create(srsran_hip_sch_enc_t** h);
SRSRAN_API void srsran_hip_sch_enc_free(srsran_hip_sch_enc_t* h);
SRSRAN_API int  srsran_hip_sch_encode(srsran_hip_sch_enc_t* h, const uint8_t* d_data, const srsran_hip_tb_t* tbs, uint32_t n_tb, uint8_t* d_e_bits,
                                      void* stream);

/* ---- the transmit side's entry: encode_tb (sch.c:239-368) as srsran_dlsch_encode2 (:625-658) reaches it, one transport block on HOST
 * buffers.  data: tbs / 8 payload bytes, or NULL for a retransmission of what the soft buffer holds; e_bits: byte-packed, nof_e_bits bits
 * from bit 0 (the unused bits of the last byte as srsran_rm_turbo_tx_lut leaves them).  softbuffer (softbuffer.h:49-53): the rows keep each
 * code block's payload slice between calls (the reference keeps its coded bits there; the contents are private to the encoder either way).
 * tests/ref_link/tb_bind.c shows the reference-side binding. */
typedef struct SRSRAN_API {
  uint32_t  max_cb;
  uint32_t  max_cb_size;
  uint8_t** buffer_b;
} srsran_softbuffer_tx_t;
SRSRAN_API int srsran_hip_encode_tb(srsran_softbuffer_tx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits,
                                    uint8_t* data, uint8_t* e_bits);

#ifdef __cplusplus
}
#endif
#endif
