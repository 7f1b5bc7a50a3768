/*
 * phy_chan_abi.h -- one device call per GRANT: the stages the reference runs between the resource grid and the transport block
 * stay on the device.
 *
 * Reference call sites these entry points are bound at (tests/ref_link/chan_bind.c shows the binding a maintainer adds):
 *   srsran_pusch_decode   lib/src/phy/phch/pusch.c:358-478   extract REs -> srsran_predecoding_single -> srsran_dft_precoding ->
 *                                                            srsran_demod_soft_demodulate_{s,b} -> srsran_sequence_pusch_apply_{s,c} ->
 *                                                            srsran_ulsch_decode (sch.c:1121: channel de-interleaver, decode_tb)
 *   srsran_pdsch_decode   lib/src/phy/phch/pdsch.c:788-946   (per codeword, :662-760) demodulate -> srsran_sequence_pdsch_apply_{s,c} ->
 *                                                            srsran_dlsch_decode2 (sch.c:579); single port: srsran_predecoding_single in front
 *   srsran_pdsch_encode   lib/src/phy/phch/pdsch.c:1017-1144 (per codeword, :949-1015) srsran_dlsch_encode2 -> srsran_sequence_pdsch_apply_pack ->
 *                                                            srsran_mod_modulate_bytes (-> power scaling, :1119)
 *   srsran_ulsch_encode   lib/src/phy/phch/sch.c:1194-1340   (without UCI) encode_tb -> channel interleaver
 * Through the per-stage handle API a grant costs four host <-> device round trips; here it costs one: symbols and channel estimates go up
 * (the kernels read them from the pinned staging image), the payload and the verdict come down.  Soft-buffer handling (HARQ combining
 * across calls, stored code blocks) is srsran_hip_decode_tb_cb's (phy_sch_abi.h).
 *
 * What is NOT taken here and stays with the caller: resource (de)mapping other than the PUSCH's rectangular one, MIMO layer mapping /
 * precoding (more than one port), UCI multiplexing (a grant with ACK / RI / CQI bits goes down the reference's own path), EVM measurement.
 */
#ifndef SRSRAN_AMD_PHY_CHAN_ABI_H
#define SRSRAN_AMD_PHY_CHAN_ABI_H

#include "srsran_amd/phy_modem_abi.h"
#include "srsran_amd/phy_sch_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the transport-block part every grant shares (srsran_ra_tb_t, ra.h:60-70, + what pusch.c / pdsch.c take from their cfg) */
typedef struct SRSRAN_API {
  uint32_t mod;                /* srsran_mod_t */
  uint32_t tbs;                /* transport block size in bits */
  uint32_t rv;                 /* redundancy version */
  uint32_t nof_re;             /* modulation symbols of the grant; nof_bits = nof_re * Qm */
  uint32_t seed;               /* c_init of the scrambling sequence: srsran_hip_sequence_pusch_seed / _pdsch_seed */
  uint32_t max_nof_iterations; /* turbo half iterations (srsran_sch_set_max_noi) */
  uint32_t llr_is_8bit;        /* q->llr_is_8bit: 8-bit soft bits, de-matcher and decoders */
  uint32_t nl;                 /* layers per codeword for the rate matcher's granularity Qm * Nl (sch.c:590,632: 2 when nof_layers != nof_tb); 0 = 1 */
} srsran_hip_grant_tb_t;

typedef struct SRSRAN_API {
  int32_t crc_ok;               /* 1 when every code block and the transport block passed (out->crc) */
  float   avg_iterations_block; /* q->ul_sch.avg_iterations / srsran_sch_last_noi */
  float   epre;                 /* PUSCH with meas_epre: average power of the extracted REs (linear; the caller converts to dB), else NAN */
} srsran_hip_grant_res_t;

/* ---- PUSCH receive (srsran_pusch_decode without UCI): sf_symbols / ce are the subframe's resource grids as srsran_ofdm_rx_sf and the channel
 * estimator leave them (14 or 12 symbols x 12 nof_prb), HOST memory; the grant's REs are taken from both as pusch.c:48-104 does. */
typedef struct SRSRAN_API {
  srsran_hip_grant_tb_t tb;
  uint32_t cell_nof_prb;   /* width of the grid */
  uint32_t cp_nsymb;       /* 7 (normal) or 6 (extended) symbols per slot */
  uint32_t n_prb_tilde[2]; /* first PRB of the allocation in each slot */
  uint32_t L_prb;          /* PRBs: transform precoding over 12 L_prb sub-carriers (srsran_dft_precoding_valid_prb) */
  uint32_t shortened;      /* last symbol taken by SRS */
  float    noise_estimate; /* channel->noise_estimate */
  uint32_t meas_epre;
} srsran_hip_pusch_rx_t;
SRSRAN_API int srsran_hip_pusch_decode(const srsran_hip_pusch_rx_t* g, const cf_t* sf_symbols, const cf_t* ce, srsran_softbuffer_rx_t* softbuffer,
                                       uint8_t* data, srsran_hip_grant_res_t* res);
/* the grants of one TTI (the loop of srsenb/src/phy/lte/cc_worker.cc:359-371 over the UEs with a grant) in ONE call: one launch per stage and
 * block size over the code blocks of all grants.  Arrays of n entries; grants whose tb.llr_is_8bit / tb.max_nof_iterations differ are
 * decoded in separate passes.  Returns SRSRAN_SUCCESS when every grant was processed (res[i].crc_ok tells its outcome). */
SRSRAN_API int srsran_hip_pusch_decode_multi(uint32_t n, const srsran_hip_pusch_rx_t* g, const cf_t* const* sf_symbols, const cf_t* const* ce,
                                             srsran_softbuffer_rx_t* const* softbuffers, uint8_t* const* data, srsran_hip_grant_res_t* res);

/* ---- PDSCH receive, one codeword: `symbols` are the grant's nof_re extracted REs (srsran_pdsch_get) of HOST memory.  ce != NULL: single
 * port, single receive antenna -- the zero-forcing / MMSE equaliser srsran_predecoding_single(symbols, ce, ., NULL, nof_re, scaling,
 * noise_estimate) runs on the device; ce == NULL: `symbols` are already equalised and layer-demapped (q->d[codeword], pdsch.c:880-899). */
typedef struct SRSRAN_API {
  srsran_hip_grant_tb_t tb;
  float scaling;        /* pdsch_scaling (rho_a), 1.0f without power allocation */
  float noise_estimate; /* 0 for SRSRAN_MIMO_DECODER_ZF */
} srsran_hip_pdsch_rx_t;
SRSRAN_API int srsran_hip_pdsch_decode(const srsran_hip_pdsch_rx_t* g, const cf_t* symbols, const cf_t* ce, srsran_softbuffer_rx_t* softbuffer,
                                       uint8_t* data, srsran_hip_grant_res_t* res);
/* the same, and the intermediate results the reference leaves in the PDSCH object where its callers can see them (lib/test/phy/phy_dl_test.c:253-298 compares
 * both with the transmitter's): d_out (or NULL) <- the nof_re equalised symbols (q->d[cw]; only written when ce != NULL -- otherwise they are `symbols`),
 * e_out (or NULL) <- the nof_re * Qm descrambled soft bits (q->e[cw]: int16, int8 with llr_is_8bit).  Each costs a device -> host copy. */
SRSRAN_API int srsran_hip_pdsch_decode_dbg(const srsran_hip_pdsch_rx_t* g, const cf_t* symbols, const cf_t* ce, srsran_softbuffer_rx_t* softbuffer,
                                           uint8_t* data, srsran_hip_grant_res_t* res, cf_t* d_out, void* e_out);

/* ---- PDSCH transmit, one codeword (srsran_pdsch_codeword_encode, pdsch.c:949-1015, + the scaling of :1116-1120): payload bytes ->
 * CRC24A, segmentation, CRC24B, turbo coding, rate matching (encode_tb, sch.c:238-368) -> scrambling -> constellation points x scaling.
 * data == NULL: a retransmission of what the soft buffer holds.  symbols: nof_re points, HOST memory. */
typedef struct SRSRAN_API {
  srsran_hip_grant_tb_t tb; /* max_nof_iterations, llr_is_8bit unused */
  float scaling;            /* 1.0f: none */
} srsran_hip_pdsch_tx_t;
SRSRAN_API int srsran_hip_pdsch_encode(const srsran_hip_pdsch_tx_t* g, srsran_softbuffer_tx_t* softbuffer, uint8_t* data, cf_t* symbols);
/* the same, and e_out (or NULL) <- the scrambled, byte-packed coded bits the reference leaves in q->e[cw] (pdsch.c:1005-1012), nof_re * Qm bits rounded
 * up to whole bytes */
/* the codewords of one TTI in ONE call (the loop over the scheduled UEs of srsenb/src/phy/lte/cc_worker.cc encode_pdsch, each of which ends in
 * srsran_pdsch_encode, pdsch.c:1017): arrays of n entries, one coding launch over the code blocks of all of them and one scrambling + modulation launch. */
SRSRAN_API int srsran_hip_pdsch_encode_multi(uint32_t n, const srsran_hip_pdsch_tx_t* g, srsran_softbuffer_tx_t* const* softbuffers, uint8_t* const* data,
                                             cf_t* const* symbols);
SRSRAN_API int srsran_hip_pdsch_encode_dbg(const srsran_hip_pdsch_tx_t* g, srsran_softbuffer_tx_t* softbuffer, uint8_t* data, cf_t* symbols, uint8_t* e_out);

/* ---- UL-SCH transmit without UCI (srsran_ulsch_encode, sch.c:1194-1340, with no ACK / RI / CQI configured): encode_tb -> channel
 * interleaver of 36.212 5.2.2.8 over nof_symb columns.  q_bits: nof_bits = nof_re * Qm bits, byte packed (what pusch.c:322 scrambles next). */
SRSRAN_API int srsran_hip_ulsch_encode(const srsran_hip_grant_tb_t* tb, uint32_t nof_symb, srsran_softbuffer_tx_t* softbuffer, uint8_t* data, uint8_t* q_bits);

/* ---- warm start.  The first grant of a process / of a worker thread otherwise pays for loading the kernels' device code, creating the thread's staging
 * context (stream, pinned and device images, decoder / encoder objects, transform plans) and building rate-matching tables: 20-28 ms where a warm
 * call takes 0.1-0.4 ms.  srsran_hip_warmup(n) builds every rate-matching table and prepares n staging contexts by running real grants (the
 * largest of a 100-PRB cell, a one-block and a scalar-decoder one; receive and transmit side; 16- and 8-bit soft bits) on short-lived threads; a worker
 * thread adopts a prepared context at its first call.  srsran_rm_turbo_gentables() -- which srsran_sch_init calls (sch.c:166) -- does the same
 * for one worker, so an application that does nothing gets a warm first subframe on one thread; srsenb's pool of nof_phy_threads workers
 * wants srsran_hip_warmup(nof_phy_threads) once after its objects are created.  Idempotent; returns SRSRAN_ERROR without a device. */
SRSRAN_API int srsran_hip_warmup(uint32_t nof_workers);

/* modulator alone: srsran_mod_modulate_bytes (mod.c:135-166) of byte-packed bits with the tables of lte_tables.c, optional scrambling in front
 * (srsran_sequence_apply_pack) and scaling behind; HOST buffers.  Returns the number of symbols or -1. */
SRSRAN_API int srsran_hip_modulate_bytes(uint32_t mod, const uint8_t* bits, cf_t* symbols, uint32_t nbits, uint32_t seed, uint32_t scramble, float scaling);

#ifdef __cplusplus
}
#endif
#endif
