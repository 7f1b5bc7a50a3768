/*
 * oracle.h -- CPU restatement of the reference's PHY DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as the
 * checker / the reported CPU baseline.  The HIP library (srslte_amd/csrc) never links or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Parity pinning:
 *   - turbo / QPP / LDPC: pinned bit-exactly against the reference's own compiled sources
 *     (oracle/_ref/libsrsran_ref.so, built by oracle/Makefile) and against the reference's golden
 *     vectors (turbodecoder_test.h K=504 known answer; examplesBG1/BG2.dat) -- tests/test_oracle_*.py
 *   - PSS / SSS: pinned to the recorded air captures the reference's own tests hold (phch/test/signal.1.92M.dat,
 *     signal.1.92M.amar.dat, signal.10M.dat with the cell ids of phch/test/CMakeLists.txt:433,439-442: 150, 1, 150):
 *     tests/golden/sync_captures.npz, tests/test_oracle_golden.py::test_sync_oracle_on_reference_captures; all 504 SSS
 *     sequences bit-equal to gen_sss.c (part of oracle/_ref).
 *   - OFDM / DFT: the reference's dft_fftw.c needs FFTW3, absent from this image -> ofdm.c / dft_fftw.c are unbuildable here.
 *     Pinned through the reference's OWN callers instead: tests/ref_link links the reference's unmodified test programs against this
 *     oracle (tests/ref_link/oracle_shim.c) and they must reach the known answers the reference holds for srsran_ofdm_rx_sf -- the MIB of
 *     signal.1.92M.dat, CFI 2 of signal.10M.dat, the DCI and PDSCH CRC of signal.1.92M.amar.dat, the PMCH CRC of the 100-PRB MBSFN
 *     subframe -- plus ofdm_test's loop-back criterion (ofdm_test.c:176) for 6 ... 110 PRB: tests/test_ref_link_oracle.py (CPU).
 *     FFT VALUES beyond 1e-4 stay unpinned, exactly as in the reference (FFTW is an unpinned system library).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- LTE turbo code (TS 36.212 5.1.3) ---------------- */

/* cbsegm.c:119-140 */
int      orc_tc_cb_index(uint32_t long_cb); /* smallest table index with size >= long_cb, -1 if none */
int      orc_tc_cb_size(uint32_t index);    /* -1 if out of range */
/* tc_interl_lte.c:61-109 ; win = 1, 8, 16 or 32 */
int      orc_qpp_gen(uint32_t long_cb, uint32_t win, uint16_t* forward, uint16_t* reverse);
/* turbodecoder.c:381-393 (AVX2 host) */
uint32_t orc_tdec_autoimp_subblocks(uint32_t long_cb);
/* turbodecoder.c:410-424 (AVX2 host) */
uint32_t orc_tdec_autoimp_subblocks_8bit(uint32_t long_cb);

/* implementation selector, values follow srsran_tdec_impl_type_t (turbodecoder_impl.h:28-38) */
enum {
  ORC_TDEC_AUTO = 0,
  ORC_TDEC_GENERIC = 1,
  ORC_TDEC_SSE_WINDOW = 3,
  ORC_TDEC_AVX_WINDOW = 5,
  ORC_TDEC_SSE8_WINDOW = 6,
  ORC_TDEC_AVX8_WINDOW = 7
};

/* srsran_tdec_run_all (turbodecoder.c:536-549) for 16-bit LLRs.
 *   input    : natural order [s0 p0 p0' s1 ...][12 tail]  (3K+12)  when sb_layout == 0
 *              rm_turbo sub-block layout (3(K+32)+12)          when sb_layout == 1 (window impls only)
 *   output   : K/8 bytes, MSB first
 *   snap     : optional, nof_iterations*K int16: the SISO output vector (ext1 on even, ext2 on odd
 *              n_iter) after every half iteration, in the decoder's internal lane layout
 *   dec_llr  : optional, K int16: the LLRs the hard decision is taken from (app1 / ext1), natural order
 * returns 0, or -1 on invalid arguments */
int orc_tdec_run_all(const int16_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb,
                     int impl, int sb_layout, int16_t* snap, int16_t* dec_llr);

/* srsran_tdec_run_all_8bit (turbodecoder.c:560-577) for int8 LLRs: AUTO (32 sub-blocks K%32==0 && K>2048,
 * 16 sub-blocks K%16==0 && K>800, else widened to the 16-bit decoders, turbodecoder.c:410-478),
 * ORC_TDEC_SSE8_WINDOW (16) or ORC_TDEC_AVX8_WINDOW (32).  dec_llr: optional K decision LLRs (int16 holder). */
int orc_tdec_run_all_8bit(const int8_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb,
                          int impl, int sb_layout, int16_t* dec_llr);

/* turbo rate de-matching, receive side (rm_turbo.c:175-273,390-478).  nof_sb: 0 = natural buffer [d0 d1 d2] x (K+4);
 * 8 / 16 / 32 = the sub-block layout of the window decoders (syst @0, parity0 @K+32, parity1 @2(K+32), tail @3(K+32)).
 * table: 3K+12 entries.  rx: output (3K+12, or 3(K+32)+12 for a sub-block layout) is accumulated into (HARQ combining). */
int orc_rm_turbo_deinter(uint16_t* table, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb);
int orc_rm_turbo_rx(const int16_t* input, int16_t* output, uint32_t in_len, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb);
int orc_rm_turbo_rx_8bit(const int8_t* input, int8_t* output, uint32_t in_len, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb);

/* srsran_cbsegm (cbsegm.c:62-117) and decode_tb / decode_tb_cb (sch.c:370-560) for one transport block: rate de-matching
 * into per-code-block soft buffers (18600 int16 each), turbo half iterations with CRC early stop, transport-block CRC.
 * The loop itself cannot be pinned against the compiled reference (sch.c drags in the whole PHY channel layer); every
 * step it is made of is. */
int orc_cbsegm(uint32_t tbs, uint32_t* C, uint32_t* K1, uint32_t* K2, uint32_t* C1, uint32_t* C2, uint32_t* F);
int orc_sch_decode_tb(uint32_t tbs, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, const int16_t* e_bits, int16_t* softbuf,
                      uint8_t* cb_crc, uint8_t* cb_data, uint32_t max_iterations, uint8_t* data, float* avg_iterations);
int orc_sch_decode_tb_8bit(uint32_t tbs, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, const int8_t* e_bits, int8_t* softbuf,
                      uint8_t* cb_crc, uint8_t* cb_data, uint32_t max_iterations, uint8_t* data, float* avg_iterations);

/* turbocoder.c:69-160 (bit-per-byte in, 3K+12 bit-per-byte out, natural order) */
int orc_tcod_encode(const uint8_t* input, uint8_t* output, uint32_t long_cb);

/* ---------------- NR LDPC (TS 38.212 5.3.2) ---------------- */

typedef struct {
  int      bg;   /* 0 = BG1, 1 = BG2 */
  uint16_t ls;   /* lifting size Z */
  int      bgN, bgM, bgK;
  int      nof_edges;
  /* per layer (check row): first edge, number of edges */
  uint16_t row_start[47];
  /* per edge: variable node and shift (V mod Z) */
  uint8_t  col[320];
  uint16_t shift[320];
} orc_ldpc_graph_t;

/* base_graph.c:4467-4503 + base_graph.h:109-113; returns -1 for an invalid lifting size */
int orc_ldpc_graph(orc_ldpc_graph_t* g, int bg, uint16_t ls);
int orc_ldpc_ls_index(uint16_t ls); /* 0..7 or -1 */

/* srsran_ldpc_decoder_decode_c (ldpc_decoder.c:44-104,657-685 with ldpc_dec_c.c), int8 layered min-sum.
 *   llrs        : N-2Z ... at least cdwd_rm_length values (LLR > 0 <=> bit 0)
 *   message     : liftK bytes, one bit per byte
 *   crc_poly/crc_order : if crc_order > 0 the CRC (crc.c) is checked after each iteration for early
 *                 stop: returns the iteration count, 0 if it never matched.  crc_order == 0: no CRC,
 *                 returns max_nof_iter.
 *   soft_out    : optional liftN int8 a-posteriori soft bits after the last executed iteration */
int orc_ldpc_decode_c(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const int8_t* llrs,
                      uint8_t* message, uint32_t cdwd_rm_length, uint32_t crc_poly, int crc_order,
                      int8_t* soft_out);

/* the flooded schedule (SRSRAN_LDPC_DECODER_C_FLOOD and its SIMD siblings; ldpc_decoder.c:105-160, ldpc_dec_c_flood.c): same
 * arguments and return convention; runs 2 * max_nof_iter iterations */
int orc_ldpc_decode_c_flood(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const int8_t* llrs,
                            uint8_t* message, uint32_t cdwd_rm_length, uint32_t crc_poly, int crc_order, int8_t* soft_out);

/* srsran_ldpc_decoder_decode_s / _decode_f (ldpc_dec_s.c, ldpc_dec_f.c), fixed number of iterations (no CRC entry
 * point exists for these types, ldpc_decoder.h:111-180).  soft_out: optional liftN a-posteriori values */
int orc_ldpc_decode_s(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const int16_t* llrs,
                      uint8_t* message, uint32_t cdwd_rm_length, int16_t* soft_out);
int orc_ldpc_decode_f(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const float* llrs,
                      uint8_t* message, uint32_t cdwd_rm_length, float* soft_out);

/* ldpc_enc_c.c / ldpc_encoder.c: systematic encoder, bit per byte, filler bits (value 254) allowed.
 * output: N-2Z bits (cdwd_rm_length = full) */
int orc_ldpc_encode(const orc_ldpc_graph_t* g, const uint8_t* message, uint8_t* codeword);
/* srsran_ldpc_encoder_encode_rm (ldpc_encoder.c:55-95,584-591): raw systematic part, parity of the rows cdwd_rm_length needs */
int orc_ldpc_encode_rm(const orc_ldpc_graph_t* g, const uint8_t* input, uint8_t* output, uint32_t cdwd_rm_length);
/* srsran_ldpc_rm_tx / srsran_ldpc_rm_rx_{c,s,f} (ldpc_rm.c); type: 0 int8, 1 int16, 2 float; Qm = bits per symbol */
int orc_ldpc_rm_tx(const uint8_t* input, uint8_t* output, uint32_t E, int bg, uint32_t ls, uint32_t rv, uint32_t Qm, uint32_t Nref);
int orc_ldpc_rm_rx(int type, const void* input, void* output, uint32_t E, uint32_t F, int bg, uint32_t ls, uint32_t rv, uint32_t Qm,
                   uint32_t Nref);

/* ---------------- NR shared channel, transport-block loop (sch_nr.c, cbsegm.c:159-285): orc_sch_nr.c ---------------- */
typedef struct { /* srsran_sch_nr_tb_info_t, sch_nr.h */
  uint32_t bg, Qm, A, L_tb, L_cb, B, Bp, Kp, Kr, F, Z, G, Nl, Nref, C;
} orc_nr_tb_info_t;
int      orc_sch_nr_select_basegraph(uint32_t tbs, double R);
int      orc_sch_nr_tb_info(uint32_t tbs, double R, uint32_t Qm, uint32_t nof_bits, uint32_t N_L, uint32_t Nref, orc_nr_tb_info_t* cfg);
uint32_t orc_sch_nr_get_E(const orc_nr_tb_info_t* cfg, uint32_t j);
int      orc_sch_nr_encode_tb(const orc_nr_tb_info_t* cfg, uint32_t rv, const uint8_t* data, uint8_t* e_bits);
int      orc_sch_nr_decode_tb(const orc_nr_tb_info_t* cfg, uint32_t rv, float scaling_fctr, int max_nof_iter, const int8_t* e_bits,
                              int8_t* softbuf, uint32_t sb_stride, uint8_t* cb_crc, uint8_t* cb_data, uint32_t data_stride, uint8_t* payload,
                              int* crc_ok, float* avg_iter);

/* crc.c:  bit-per-byte CRC as srsran_crc_checksum (no reversal) */
uint32_t orc_crc_bits(uint32_t poly, int order, const uint8_t* bits, int len);

/* ---------------- OFDM / DFT ---------------- */

typedef struct {
  uint32_t nof_prb;
  uint32_t symbol_sz;        /* 0 -> orc_symbol_sz(nof_prb) (non-standard rates, phy_common.c:361-385) */
  int      cp_ext;           /* 0 normal, 1 extended */
  int      normalize;
  float    freq_shift_f;
  float    rx_window_offset;
  int      keep_dc;
  int      mbsfn_region;     /* 0: normal subframe; 1, 2: MBSFN subframe with this non-MBSFN region (needs cp_ext) */
} orc_ofdm_cfg_t;

int  orc_symbol_sz(uint32_t nof_prb);            /* phy_common.c:361-385, default (non standard) rates */
int  orc_symbol_sz_power2(uint32_t nof_prb);     /* phy_common.c:342-359 */
int  orc_cp_len(uint32_t symbol_sz, int c);      /* SRSRAN_CP_LEN, phy_common.h:125 */
/* srsran_ofdm_rx_sf (ofdm.c:453-466).  in: 15*N cf (interleaved re,im floats; NOT modified),
 * out: nsymb*2*12*nof_prb cf.  Arithmetic in double, rounded to float at the end. */
int  orc_ofdm_rx_sf(const orc_ofdm_cfg_t* cfg, const float* in, float* out);
/* srsran_ofdm_tx_sf (ofdm.c:562-576) */
int  orc_ofdm_tx_sf(const orc_ofdm_cfg_t* cfg, const float* in, float* out);
/* srsran_dft_run_c (dft_fftw.c:297-354): forward (dir=0) / backward (dir=1), options mirror,dc,norm */
void orc_dft_c(const float* in, float* out, int n, int backward, int mirror, int dc, int norm);
/* fast float FFT used by the CPU-baseline timing leg (same math as orc_dft_c without options) */
void orc_fft_f32(const float* in, float* out, int n, int backward);

/* ---------------- PSS / SSS ---------------- */
/* pss.c:341-368: 62 complex values */
int  orc_pss_generate(float* signal, uint32_t N_id_2);
/* gen_sss.c:55-163: two 62-float sequences for subframes 0 and 5 */
int  orc_sss_generate(float* sf0, float* sf5, uint32_t cell_id);
/* pss.c:31-62,446-534: correlate `frame_size` samples with the time-domain PSS replica of N_id_2,
 * |.|^2, argmax.  corr_out (optional) receives frame_size+fft_size-1 floats.  Returns peak index,
 * peak value and PSR (pss.c:408-437) */
int  orc_pss_find(const float* input, uint32_t frame_size, uint32_t fft_size, uint32_t N_id_2,
                  float* corr_out, float* peak_value, float* psr);
/* pieces of orc_pss_find for an FFT-based port: the time-domain replica (fft_size complex) and the PSR rule on a power vector
 * that holds conv_output_len + 2 entries (zero behind the computed ones) */
int   orc_pss_time_replica(float* out, uint32_t N_id_2, uint32_t fft_size);
float orc_peak_sidelobe(const float* avg, uint32_t corr_peak_pos, uint32_t conv_output_len);
/* find_sss.c:99-192 partial + sss.c:128-156; input points at the start of the SSS symbol (N samples) */
int  orc_sss_m0m1_partial(const float* sss_symbol, uint32_t fft_size, uint32_t N_id_2, uint32_t* m0,
                          uint32_t* m1, int* n_id_1, int* sf_idx);
/* general form: M = 1 (full), 3 (partial) as srsran_sss_m0m1_partial; M = 0: srsran_sss_m0m1_diff */
int  orc_sss_m0m1(const float* sss_symbol, uint32_t fft_size, uint32_t N_id_2, uint32_t M, uint32_t* m0,
                  float* m0_value, uint32_t* m1, float* m1_value, int* n_id_1, int* sf_idx);

/* helpers behind srsran_sync_find: cexptab.c / cfo.c:101-113 / cp.c:60-79 / pss.c:590-640 / sync.c:451-508.
 * cfo + cp_synch are pinned against oracle/_ref (those files build without FFTW) */
void     orc_cexptab_gen(float* x, float freq, uint32_t len);
void     orc_cfo_correct(const float* in, float* out, float freq, uint32_t n);
uint32_t orc_cp_synch(const float* in, uint32_t N, uint32_t max_offset, uint32_t nof_symbols, uint32_t cp_len, float* corr);
int      orc_pss_filter(const float* in, float* out, uint32_t N, uint32_t N_id_2, float* ce);
float    orc_pss_cfo_compute(const float* pss_recv, uint32_t N, uint32_t N_id_2);
int      orc_detect_cp(const float* in, uint32_t peak_pos, uint32_t N, float* m);

/* ---------------- soft demodulation + descrambling (orc_modem.c) ---------------- */
/* demod_soft.c: mod = srsran_mod_t (0 BPSK .. 4 256QAM), x = n complex symbols (re,im floats) */
int      orc_demod_soft_s(int mod, const float* x, int16_t* llr, int n);
int      orc_demod_soft_b(int mod, const float* x, int8_t* llr, int n);
int      orc_demod_soft_f(int mod, const float* x, float* llr, int n);
/* sequence.c: c(0..len-1) of the length-31 Gold sequence with c_init = seed; scratch: len bytes */
void     orc_sequence_bits(uint32_t seed, uint8_t* c, uint32_t len);
void     orc_sequence_apply_s(const int16_t* in, int16_t* out, uint32_t len, uint32_t seed, uint8_t* scratch);
void     orc_sequence_apply_c(const int8_t* in, int8_t* out, uint32_t len, uint32_t seed, uint8_t* scratch);
void     orc_sequence_apply_f(const float* in, float* out, uint32_t len, uint32_t seed, uint8_t* scratch);
uint32_t orc_sequence_pdsch_seed(uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id);
uint32_t orc_sequence_pusch_seed(uint16_t rnti, uint32_t nslot, uint32_t cell_id);
/* srsran_predecoding_single (mimo/precoding.c:357-392): single-antenna ZF / MMSE equaliser, double arithmetic */
int      orc_predecoding_single(const float* y, const float* h, float* x, float* csi, int n, float scaling, float noise_estimate);
/* lte_tables.c:30-181 / modem_table.c: constellation of srsran_mod_t `mod`, 2^Qm points (re, im), index = bits MSB first */
int      orc_mod_table(int mod, float* out);
/* sequence.c:609-650 + mod.c:135-166 (+ vector scaling): packed bits -> [scrambled with c_init = seed] -> points x scaling */
int      orc_modulate_bytes(int mod, const uint8_t* bits, float* out, uint32_t nbits, uint32_t seed, int scramble, float scaling);
/* sch.c:660-681 without RI bits: lut[q position] = g position */
int      orc_ulsch_interleaver_lut(uint32_t nof_sym, uint32_t Qm, uint32_t cols, uint32_t* lut);

#ifdef __cplusplus
}
#endif
#endif
