/*
 * phy_abi.h -- drop-in C ABI of the MI355X PHY DSP engine (libsrsran_phy_hip.so).
 *
 * These are the entry points an srsRAN 21.04 build binds for the hot path.  Names, argument
 * meaning, return codes and -- because the handles are caller-allocated and embedded by value in
 * srsran_ue_dl_t / srsran_enb_ul_t / srsran_sch_t / srsran_sch_nr_t -- struct sizes and field
 * offsets are those of the reference headers cited at each block (paths relative to the
 * reference tree).  tests/test_host_cpu.py (test_struct_layout_matches_reference / _recorded_reference) checks sizes/offsets against the reference headers.
 *
 * Handles keep their device state behind the pointer fields the reference already has
 * (srsran_dft_plan_t.p, srsran_tdec_t.dec16_hdlr[0], srsran_ldpc_decoder_t.ptr, srsran_ofdm_t.tmp):
 * no field was added or re-typed.
 *
 * I/O buffers are HOST memory owned by the caller exactly as in the reference; every run call is
 * synchronous (H2D, kernels, D2H on the handle's stream, then stream sync).  For device-resident
 * batched operation see phy_batch.h.
 */
#ifndef SRSRAN_AMD_PHY_ABI_H
#define SRSRAN_AMD_PHY_ABI_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
#include <complex>
typedef std::complex<float> cf_t; /* lib/include/srsran/config.h:67 */
extern "C" {
#else
#include <complex.h>
typedef _Complex float cf_t;
#endif

#define SRSRAN_API __attribute__((visibility("default")))

/* lib/include/srsran/config.h:57-64 */
#define SRSRAN_SUCCESS 0
#define SRSRAN_ERROR -1
#define SRSRAN_ERROR_INVALID_INPUTS -2

/* ------------------------------------------------------------------------------------------------
 * DFT  (lib/include/srsran/phy/dft/dft.h:50-125, lib/src/phy/dft/dft_fftw.c)
 * ---------------------------------------------------------------------------------------------- */
typedef enum { SRSRAN_DFT_COMPLEX, SRSRAN_REAL } srsran_dft_mode_t;
typedef enum { SRSRAN_DFT_FORWARD, SRSRAN_DFT_BACKWARD } srsran_dft_dir_t;

typedef struct SRSRAN_API {
  int               init_size;
  int               size;
  void*             in;  /* host staging buffer (owned unless guru) */
  void*             out; /* host staging buffer (owned unless guru) */
  void*             p;   /* -> device plan context (reference: fftwf_plan) */
  bool              is_guru;
  bool              forward;
  bool              mirror;
  bool              db;
  bool              norm;
  bool              dc;
  srsran_dft_dir_t  dir;
  srsran_dft_mode_t mode;
} srsran_dft_plan_t;

SRSRAN_API int  srsran_dft_plan(srsran_dft_plan_t* plan, int dft_points, srsran_dft_dir_t dir, srsran_dft_mode_t type);
/* real <-> half-complex (FFTW_R2HC / FFTW_HC2R layout), dft.h:76,88,121 */
SRSRAN_API int  srsran_dft_plan_r(srsran_dft_plan_t* plan, int dft_points, srsran_dft_dir_t dir);
SRSRAN_API int  srsran_dft_replan_r(srsran_dft_plan_t* plan, int new_dft_points);
SRSRAN_API void srsran_dft_run_r(srsran_dft_plan_t* plan, const float* in, float* out);
SRSRAN_API int  srsran_dft_plan_c(srsran_dft_plan_t* plan, int dft_points, srsran_dft_dir_t dir);
SRSRAN_API int  srsran_dft_plan_guru_c(srsran_dft_plan_t* plan, int dft_points, srsran_dft_dir_t dir, cf_t* in_buffer,
                                       cf_t* out_buffer, int istride, int ostride, int how_many, int idist, int odist);
SRSRAN_API int  srsran_dft_replan(srsran_dft_plan_t* plan, const int new_dft_points);
SRSRAN_API int  srsran_dft_replan_c(srsran_dft_plan_t* plan, int new_dft_points);
SRSRAN_API int  srsran_dft_replan_guru_c(srsran_dft_plan_t* plan, const int new_dft_points, cf_t* in_buffer,
                                         cf_t* out_buffer, int istride, int ostride, int how_many, int idist, int odist);
SRSRAN_API void srsran_dft_plan_free(srsran_dft_plan_t* plan);
SRSRAN_API void srsran_dft_plan_set_mirror(srsran_dft_plan_t* plan, bool val);
SRSRAN_API void srsran_dft_plan_set_db(srsran_dft_plan_t* plan, bool val);
SRSRAN_API void srsran_dft_plan_set_norm(srsran_dft_plan_t* plan, bool val);
SRSRAN_API void srsran_dft_plan_set_dc(srsran_dft_plan_t* plan, bool val);
SRSRAN_API void srsran_dft_run(srsran_dft_plan_t* plan, const void* in, void* out);
SRSRAN_API void srsran_dft_run_c_zerocopy(srsran_dft_plan_t* plan, const cf_t* in, cf_t* out);
SRSRAN_API void srsran_dft_run_c(srsran_dft_plan_t* plan, const cf_t* in, cf_t* out);
SRSRAN_API void srsran_dft_run_guru_c(srsran_dft_plan_t* plan);

/* DFT transform precoding (SC-FDMA), lib/include/srsran/phy/dft/dft_precoding.h:39-59, dft_precoding.c */
#define SRSRAN_MAX_PRB 110
#define SRSRAN_NRE 12
typedef struct SRSRAN_API {
  uint32_t          max_prb;
  srsran_dft_plan_t dft_plan[SRSRAN_MAX_PRB + 1];
} srsran_dft_precoding_t;

SRSRAN_API int      srsran_dft_precoding_init(srsran_dft_precoding_t* q, uint32_t max_prb, bool is_tx);
SRSRAN_API int      srsran_dft_precoding_init_tx(srsran_dft_precoding_t* q, uint32_t max_prb);
SRSRAN_API int      srsran_dft_precoding_init_rx(srsran_dft_precoding_t* q, uint32_t max_prb);
SRSRAN_API void     srsran_dft_precoding_free(srsran_dft_precoding_t* q);
SRSRAN_API bool     srsran_dft_precoding_valid_prb(uint32_t nof_prb);
SRSRAN_API uint32_t srsran_dft_precoding_get_valid_prb(uint32_t nof_prb);
SRSRAN_API int      srsran_dft_precoding(srsran_dft_precoding_t* q, cf_t* input, cf_t* output, uint32_t nof_prb, uint32_t nof_symbols);

/* ------------------------------------------------------------------------------------------------
 * OFDM  (lib/include/srsran/phy/dft/ofdm.h:48-142, lib/src/phy/dft/ofdm.c)
 * ---------------------------------------------------------------------------------------------- */
typedef enum { SRSRAN_CP_NORM = 0, SRSRAN_CP_EXT } srsran_cp_t; /* phy_common.h:83 */
typedef enum { SRSRAN_SF_NORM = 0, SRSRAN_SF_MBSFN } srsran_sf_t; /* phy_common.h:84 */

typedef struct SRSRAN_API {
  uint32_t    nof_prb;
  cf_t*       in_buffer;
  cf_t*       out_buffer;
  srsran_cp_t cp;
  srsran_sf_t sf_type;
  bool        normalize;
  float       freq_shift_f;
  float       rx_window_offset;
  uint32_t    symbol_sz;
  bool        keep_dc;
} srsran_ofdm_cfg_t;

typedef struct SRSRAN_API {
  srsran_ofdm_cfg_t cfg;
  srsran_dft_plan_t fft_plan;
  srsran_dft_plan_t fft_plan_sf[2];
  uint32_t          max_prb;
  uint32_t          nof_symbols;
  uint32_t          nof_guards;
  uint32_t          nof_re;
  uint32_t          slot_sz;
  uint32_t          sf_sz;
  cf_t*             tmp; /* -> device context of this OFDM object (reference: host scratch) */
  bool              mbsfn_subframe;
  uint32_t          mbsfn_guard_len;
  uint32_t          nof_symbols_mbsfn;
  uint8_t           non_mbsfn_region;
  uint32_t          window_offset_n;
  cf_t*             shift_buffer;         /* host copy of the frequency-shift table (ofdm.c:334-356) */
  cf_t*             window_offset_buffer; /* host copy of the window-offset ramp (ofdm.c:130-138) */
} srsran_ofdm_t;

SRSRAN_API int  srsran_ofdm_rx_init_cfg(srsran_ofdm_t* q, srsran_ofdm_cfg_t* cfg);
SRSRAN_API int  srsran_ofdm_tx_init_cfg(srsran_ofdm_t* q, srsran_ofdm_cfg_t* cfg);
SRSRAN_API int  srsran_ofdm_rx_init_mbsfn(srsran_ofdm_t* q, srsran_cp_t cp_type, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb);
SRSRAN_API int  srsran_ofdm_rx_init(srsran_ofdm_t* q, srsran_cp_t cp_type, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb);
SRSRAN_API int  srsran_ofdm_tx_set_prb(srsran_ofdm_t* q, srsran_cp_t cp, uint32_t nof_prb);
SRSRAN_API int  srsran_ofdm_rx_set_prb(srsran_ofdm_t* q, srsran_cp_t cp, uint32_t nof_prb);
SRSRAN_API void srsran_ofdm_rx_free(srsran_ofdm_t* q);
SRSRAN_API void srsran_ofdm_rx_sf(srsran_ofdm_t* q);
SRSRAN_API void srsran_ofdm_rx_sf_ng(srsran_ofdm_t* q, cf_t* input, cf_t* output);
SRSRAN_API int  srsran_ofdm_tx_init(srsran_ofdm_t* q, srsran_cp_t cp_type, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb);
SRSRAN_API int  srsran_ofdm_tx_init_mbsfn(srsran_ofdm_t* q, srsran_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb);
SRSRAN_API void srsran_ofdm_tx_free(srsran_ofdm_t* q);
SRSRAN_API void srsran_ofdm_tx_sf(srsran_ofdm_t* q);
SRSRAN_API int  srsran_ofdm_set_freq_shift(srsran_ofdm_t* q, float freq_shift);
SRSRAN_API void srsran_ofdm_set_normalize(srsran_ofdm_t* q, bool normalize_enable);
SRSRAN_API void srsran_ofdm_set_non_mbsfn_region(srsran_ofdm_t* q, uint8_t non_mbsfn_region);

/* phy_common.c:322-385 (the OFDM init depends on them) */
SRSRAN_API int  srsran_symbol_sz(uint32_t nof_prb);
SRSRAN_API int  srsran_symbol_sz_power2(uint32_t nof_prb);
SRSRAN_API void srsran_use_standard_symbol_size(bool enabled);

/* ------------------------------------------------------------------------------------------------
 * Turbo decoder  (lib/include/srsran/phy/fec/turbo/turbodecoder.h:63-121, tc_interl.h:36-48,
 *                 lib/include/srsran/phy/fec/cbsegm.h, lib/src/phy/fec/turbo/turbodecoder.c)
 * ---------------------------------------------------------------------------------------------- */
#define SRSRAN_TCOD_RATE 3
#define SRSRAN_TCOD_TOTALTAIL 12
#define SRSRAN_TCOD_MAX_LEN_CB 6144
#define SRSRAN_NOF_TC_CB_SIZES 188
#define SRSRAN_TDEC_EXPECT_INPUT_SB 1
#define SRSRAN_TDEC_NOF_AUTO_MODES_8 2
#define SRSRAN_TDEC_NOF_AUTO_MODES_16 3

typedef struct SRSRAN_API {
  uint16_t* forward;
  uint16_t* reverse;
  uint32_t  max_long_cb;
} srsran_tc_interl_t;

SRSRAN_API int  srsran_tc_interl_init(srsran_tc_interl_t* h, uint32_t max_long_cb);
SRSRAN_API void srsran_tc_interl_free(srsran_tc_interl_t* h);
SRSRAN_API int  srsran_tc_interl_LTE_gen(srsran_tc_interl_t* h, uint32_t long_cb);
SRSRAN_API int  srsran_tc_interl_LTE_gen_interl(srsran_tc_interl_t* h, uint32_t long_cb, uint32_t interl_win);
SRSRAN_API int  srsran_cbsegm_cbindex(uint32_t long_cb);
SRSRAN_API int  srsran_cbsegm_cbsize(uint32_t index);

/* turbodecoder_impl.h:28-38 */
typedef enum SRSRAN_API {
  SRSRAN_TDEC_AUTO = 0,
  SRSRAN_TDEC_GENERIC,
  SRSRAN_TDEC_SSE,
  SRSRAN_TDEC_SSE_WINDOW,
  SRSRAN_TDEC_NEON_WINDOW,
  SRSRAN_TDEC_AVX_WINDOW,
  SRSRAN_TDEC_SSE8_WINDOW,
  SRSRAN_TDEC_AVX8_WINDOW,
  SRSRAN_TDEC_NOF_IMP
} srsran_tdec_impl_type_t;

/* turbodecoder_impl.h:53-59; kept for layout only -- the HIP engine does not dispatch through it */
typedef struct SRSRAN_API {
  int (*tdec_init)(void** h, uint32_t max_long_cb);
  void (*tdec_free)(void* h);
  void (*tdec_dec)(void* h, int8_t* input, int8_t* app, int8_t* parity, int8_t* output, uint32_t long_cb);
  void (*tdec_extract_input)(int8_t* input, int8_t* syst, int8_t* parity0, int8_t* parity1, int8_t* app2, uint32_t long_cb);
  void (*tdec_decision_byte)(int8_t* app1, uint8_t* output, uint32_t long_cb);
} srsran_tdec_8bit_impl_t;
typedef struct SRSRAN_API {
  int (*tdec_init)(void** h, uint32_t max_long_cb);
  void (*tdec_free)(void* h);
  void (*tdec_dec)(void* h, int16_t* input, int16_t* app, int16_t* parity, int16_t* output, uint32_t long_cb);
  void (*tdec_extract_input)(int16_t* input, int16_t* syst, int16_t* parity0, int16_t* parity1, int16_t* app2, uint32_t long_cb);
  void (*tdec_decision_byte)(int16_t* app1, uint8_t* output, uint32_t long_cb);
} srsran_tdec_16bit_impl_t;

typedef enum { SRSRAN_TDEC_8, SRSRAN_TDEC_16 } srsran_tdec_llr_type_t;

typedef struct SRSRAN_API {
  uint32_t max_long_cb;

  void*                     dec8_hdlr[SRSRAN_TDEC_NOF_AUTO_MODES_8];
  void*                     dec16_hdlr[SRSRAN_TDEC_NOF_AUTO_MODES_16]; /* [0] -> device context */
  srsran_tdec_8bit_impl_t*  dec8[SRSRAN_TDEC_NOF_AUTO_MODES_8];
  srsran_tdec_16bit_impl_t* dec16[SRSRAN_TDEC_NOF_AUTO_MODES_16];
  int                       nof_blocks8[SRSRAN_TDEC_NOF_AUTO_MODES_8];
  int                       nof_blocks16[SRSRAN_TDEC_NOF_AUTO_MODES_16];

  void* app1;
  void* app2;
  void* ext1;
  void* ext2;
  void* syst0;
  void* parity0;
  void* parity1;

  void* input_conv;

  bool force_not_sb;

  srsran_tdec_impl_type_t dec_type;

  srsran_tdec_llr_type_t current_llr_type;
  uint32_t               current_dec;
  uint32_t               current_long_cb;
  uint32_t               current_inter_idx;
  int                    current_cbidx;
  srsran_tc_interl_t     interleaver[4][SRSRAN_NOF_TC_CB_SIZES];
  int                    n_iter;
} srsran_tdec_t;

SRSRAN_API int      srsran_tdec_init(srsran_tdec_t* h, uint32_t max_long_cb);
SRSRAN_API int      srsran_tdec_init_manual(srsran_tdec_t* h, uint32_t max_long_cb, srsran_tdec_impl_type_t dec_type);
SRSRAN_API void     srsran_tdec_free(srsran_tdec_t* h);
SRSRAN_API void     srsran_tdec_force_not_sb(srsran_tdec_t* h);
SRSRAN_API int      srsran_tdec_new_cb(srsran_tdec_t* h, uint32_t long_cb);
SRSRAN_API int      srsran_tdec_get_nof_iterations(srsran_tdec_t* h);
SRSRAN_API uint32_t srsran_tdec_autoimp_get_subblocks(uint32_t long_cb);
SRSRAN_API uint32_t srsran_tdec_autoimp_get_subblocks_8bit(uint32_t long_cb);
SRSRAN_API void     srsran_tdec_iteration(srsran_tdec_t* h, int16_t* input, uint8_t* output);
SRSRAN_API int      srsran_tdec_run_all(srsran_tdec_t* h, int16_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb);
SRSRAN_API void     srsran_tdec_iteration_8bit(srsran_tdec_t* h, int8_t* input, uint8_t* output);
SRSRAN_API int      srsran_tdec_run_all_8bit(srsran_tdec_t* h, int8_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb);

/* ------------------------------------------------------------------------------------------------
 * NR LDPC decoder  (lib/include/srsran/phy/fec/ldpc/ldpc_decoder.h:41-195, base_graph.h:50-113,
 *                   lib/include/srsran/phy/fec/crc.h:40-48)
 * ---------------------------------------------------------------------------------------------- */
#define MAX_CNCT 20
#define NO_CNCT 0xFFFF
#define VOID_LIFTSIZE 255
#define MAX_LIFTSIZE 384

typedef enum SRSRAN_API { BG1 = 0, BG2 } srsran_basegraph_t;

SRSRAN_API int create_compact_pcm(uint16_t* pcm, int8_t (*positions)[MAX_CNCT], srsran_basegraph_t bg, uint16_t ls);
SRSRAN_API extern const uint8_t LSindex[385];

typedef struct SRSRAN_API {
  uint64_t table[256];
  int      polynom;
  int      order;
  uint64_t crcinit;
  uint64_t crcmask;
  uint64_t crchighbit;
  uint32_t srsran_crc_out;
} srsran_crc_t;

/* The CRC object is created by the caller with the reference's own srsran_crc_init (crc.c:74-90, not part
 * of this library); srsran_ldpc_decoder_decode_crc_c only reads its `polynom` and `order` fields. */

typedef enum {
  SRSRAN_LDPC_DECODER_F = 0,
  SRSRAN_LDPC_DECODER_S,
  SRSRAN_LDPC_DECODER_C,
  SRSRAN_LDPC_DECODER_C_FLOOD,
  SRSRAN_LDPC_DECODER_C_AVX2,
  SRSRAN_LDPC_DECODER_C_AVX2_FLOOD,
  SRSRAN_LDPC_DECODER_C_AVX512,
  SRSRAN_LDPC_DECODER_C_AVX512_FLOOD,
} srsran_ldpc_decoder_type_t;

typedef struct {
  srsran_ldpc_decoder_type_t type;
  srsran_basegraph_t         bg;
  uint16_t                   ls;
  float                      scaling_fctr;
  uint32_t                   max_nof_iter;
} srsran_ldpc_decoder_args_t;

typedef struct SRSRAN_API {
  void*              ptr; /* -> device context (reference: decoder registers) */
  srsran_basegraph_t bg;
  uint16_t           ls;
  uint32_t           max_nof_iter;
  uint8_t            bgN;
  uint16_t           liftN;
  uint8_t            bgM;
  uint16_t           liftM;
  uint8_t            bgK;
  uint16_t           liftK;
  uint16_t*          pcm;
  int8_t (*var_indices)[MAX_CNCT];
  float scaling_fctr;
  void (*free)(void*);
  int (*decode_f)(void*, const float*, uint8_t*, uint32_t, srsran_crc_t*);
  int (*decode_s)(void*, const int16_t*, uint8_t*, uint32_t, srsran_crc_t*);
  int (*decode_c)(void*, const int8_t*, uint8_t*, uint32_t, srsran_crc_t*);
} srsran_ldpc_decoder_t;

SRSRAN_API int  srsran_ldpc_decoder_init(srsran_ldpc_decoder_t* q, const srsran_ldpc_decoder_args_t* args);
SRSRAN_API void srsran_ldpc_decoder_free(srsran_ldpc_decoder_t* q);
SRSRAN_API int  srsran_ldpc_decoder_decode_f(srsran_ldpc_decoder_t* q, const float* llrs, uint8_t* message, uint32_t cdwd_rm_length);
SRSRAN_API int  srsran_ldpc_decoder_decode_s(srsran_ldpc_decoder_t* q, const int16_t* llrs, uint8_t* message, uint32_t cdwd_rm_length);
SRSRAN_API int  srsran_ldpc_decoder_decode_c(srsran_ldpc_decoder_t* q, const int8_t* llrs, uint8_t* message, uint32_t cdwd_rm_length);
SRSRAN_API int  srsran_ldpc_decoder_decode_crc_c(srsran_ldpc_decoder_t* q, const int8_t* llrs, uint8_t* message, uint32_t cdwd_rm_length, srsran_crc_t* crc);

#ifdef __cplusplus
}
#endif
#endif /* SRSRAN_AMD_PHY_ABI_H */
