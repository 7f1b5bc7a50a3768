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

#ifdef __cplusplus
}
#endif
#endif
