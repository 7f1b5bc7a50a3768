/*
 * phy_sync_abi.h -- PSS / SSS part of the drop-in C ABI (libsrsran_phy_hip.so) and its batched extension.
 *
 * Struct layouts follow lib/include/srsran/phy/sync/pss.h:64-102, sss.h:44-82,
 * lib/include/srsran/phy/utils/convolution.h:37-54 and filter.h:30-38 of the reference (the PSS object
 * embeds the convolution and decimation-filter objects by value).  Round-1 coverage of this family:
 *   srsran_pss_* : init/resize/free/reset, generate, put/get_slot, set_N_id_2, set_ema_alpha, find_pss
 *   srsran_sss_* : init/resize/free, generate, put_slot, set_N_id_2, set_threshold, m0m1_partial / _diff /
 *                  _diff_coh, N_id_1, subframe
 * Not yet provided (srsran_sync_t glue of sync.c, PSS-based CFO / channel estimate helpers, decimation):
 * see DESIGN.md "gaps".
 */
#ifndef SRSRAN_AMD_PHY_SYNC_ABI_H
#define SRSRAN_AMD_PHY_SYNC_ABI_H

#include "srsran_amd/phy_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SRSRAN_SYMBOL_SZ_MAX 2048
#define SRSRAN_PSS_LEN 62
#define SRSRAN_SSS_N 31
#define SRSRAN_SSS_LEN (2 * SRSRAN_SSS_N)

/* convolution.h:37-54 (layout only: the HIP engine correlates by overlap-save, not with these plans) */
typedef struct SRSRAN_API {
  cf_t*             input_fft;
  cf_t*             filter_fft;
  cf_t*             output_fft;
  cf_t*             output_fft2;
  uint32_t          input_len;
  uint32_t          filter_len;
  uint32_t          output_len;
  uint32_t          max_input_len;
  uint32_t          max_filter_len;
  srsran_dft_plan_t input_plan;
  srsran_dft_plan_t filter_plan;
  srsran_dft_plan_t output_plan;
} srsran_conv_fft_cc_t;

/* filter.h:30-38 */
typedef struct SRSRAN_API {
  cf_t*  filter_input;
  cf_t*  downsampled_input;
  cf_t*  filter_output;
  bool   is_decimator;
  int    factor;
  int    num_taps;
  float* taps;
} srsran_filt_cc_t;

/* pss.h:64-102 */
typedef struct SRSRAN_API {
  srsran_conv_fft_cc_t conv_fft; /* conv_fft.input_fft -> device context of this PSS object */
  srsran_filt_cc_t     filter;
  int                  decimate;

  uint32_t max_frame_size;
  uint32_t max_fft_size;

  uint32_t frame_size;
  uint32_t N_id_2;
  uint32_t fft_size;
  cf_t*    pss_signal_freq_full[3];

  cf_t* pss_signal_time[3];
  cf_t* pss_signal_time_scale[3];

  cf_t   pss_signal_freq[3][SRSRAN_PSS_LEN];
  cf_t*  tmp_input;
  cf_t*  conv_output;
  float* conv_output_abs;
  float  ema_alpha;
  float* conv_output_avg;
  float  peak_value;

  bool              filter_pss_enable;
  srsran_dft_plan_t dftp_input;
  srsran_dft_plan_t idftp_input;
  cf_t              tmp_fft[SRSRAN_SYMBOL_SZ_MAX];
  cf_t              tmp_fft2[SRSRAN_SYMBOL_SZ_MAX];

  cf_t tmp_ce[SRSRAN_PSS_LEN];

  bool chest_on_filter;
} srsran_pss_t;

SRSRAN_API int  srsran_pss_init(srsran_pss_t* q, uint32_t frame_size);
SRSRAN_API int  srsran_pss_init_fft(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size);
SRSRAN_API int  srsran_pss_init_fft_offset(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size, int cfo_i);
SRSRAN_API int  srsran_pss_init_fft_offset_decim(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size, int cfo_i, int decimate);
SRSRAN_API int  srsran_pss_resize(srsran_pss_t* q, uint32_t frame_size, uint32_t fft_size, int offset);
SRSRAN_API void srsran_pss_free(srsran_pss_t* q);
SRSRAN_API void srsran_pss_reset(srsran_pss_t* q);
SRSRAN_API int  srsran_pss_generate(cf_t* signal, uint32_t N_id_2);
SRSRAN_API void srsran_pss_get_slot(cf_t* slot, cf_t* pss_signal, uint32_t nof_prb, srsran_cp_t cp);
SRSRAN_API void srsran_pss_put_slot(cf_t* pss_signal, cf_t* slot, uint32_t nof_prb, srsran_cp_t cp);
SRSRAN_API void srsran_pss_set_ema_alpha(srsran_pss_t* q, float alpha);
SRSRAN_API int  srsran_pss_set_N_id_2(srsran_pss_t* q, uint32_t N_id_2);
/* returns the peak position; *corr_peak_value = peak / side-lobe ratio (SRSRAN_PSS_RETURN_PSR, pss.h:61) */
SRSRAN_API int  srsran_pss_find_pss(srsran_pss_t* q, const cf_t* input, float* corr_peak_value);

/* sss.h:44-82 */
typedef struct SRSRAN_API {
  float z1[SRSRAN_SSS_N][SRSRAN_SSS_N];
  float c[2][SRSRAN_SSS_N];
  float s[SRSRAN_SSS_N][SRSRAN_SSS_N];
  float sd[SRSRAN_SSS_N][SRSRAN_SSS_N - 1];
} srsran_sss_fc_tables_t;

typedef struct SRSRAN_API {
  srsran_dft_plan_t dftp_input; /* dftp_input.p -> device context of this SSS object */

  uint32_t fft_size;
  uint32_t max_fft_size;

  float    corr_peak_threshold;
  uint32_t symbol_sz;
  uint32_t subframe_sz;
  uint32_t N_id_2;

  uint32_t               N_id_1_table[30][30];
  srsran_sss_fc_tables_t fc_tables[3];

  float corr_output_m0[SRSRAN_SSS_N];
  float corr_output_m1[SRSRAN_SSS_N];
} srsran_sss_t;

SRSRAN_API int      srsran_sss_init(srsran_sss_t* q, uint32_t fft_size);
SRSRAN_API int      srsran_sss_resize(srsran_sss_t* q, uint32_t fft_size);
SRSRAN_API void     srsran_sss_free(srsran_sss_t* q);
SRSRAN_API void     srsran_sss_generate(float* signal0, float* signal5, uint32_t cell_id);
SRSRAN_API void     srsran_sss_put_slot(float* sss, cf_t* symbol, uint32_t nof_prb, srsran_cp_t cp);
SRSRAN_API int      srsran_sss_set_N_id_2(srsran_sss_t* q, uint32_t N_id_2);
SRSRAN_API void     srsran_sss_set_threshold(srsran_sss_t* q, float threshold);
SRSRAN_API int      srsran_sss_m0m1_partial(srsran_sss_t* q, const cf_t* input, uint32_t M, cf_t ce[2 * SRSRAN_SSS_N], uint32_t* m0,
                                            float* m0_value, uint32_t* m1, float* m1_value);
SRSRAN_API int      srsran_sss_m0m1_diff_coh(srsran_sss_t* q, const cf_t* input, cf_t ce[2 * SRSRAN_SSS_N], uint32_t* m0,
                                             float* m0_value, uint32_t* m1, float* m1_value);
SRSRAN_API int      srsran_sss_m0m1_diff(srsran_sss_t* q, const cf_t* input, uint32_t* m0, float* m0_value, uint32_t* m1, float* m1_value);
SRSRAN_API uint32_t srsran_sss_subframe(uint32_t m0, uint32_t m1);
SRSRAN_API int      srsran_sss_N_id_1(srsran_sss_t* q, uint32_t m0, uint32_t m1, float corr);

/* ------------------------------------------------------------------------------------------------
 * Batched cell search on device-resident captures: for every capture and every N_id_2 hypothesis, what
 * srsran_sync_find does in its default configuration (sync.c:629-843 without CFO stages): PSS peak + PSR,
 * then SSS at peak - 2*(N+cp) + cp (FDD) -> m0, m1, N_id_1, subframe index.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t  peak_pos;
  float    peak_value;
  float    psr;
  int32_t  sss_available;
  uint32_t m0, m1;
  float    m0_value, m1_value;
  int32_t  N_id_1; /* -1: not found; cell id = 3 * N_id_1 + N_id_2 */
  uint32_t sf_idx;
} srsran_hip_cell_t;

typedef struct srsran_hip_cellsearch srsran_hip_cellsearch_t;

/* sss_alg: 0 = differential (SSS_DIFF), 1 = full (SSS_FULL), 3 = partial with 3 segments (SSS_PARTIAL_3), sync.h:66 */
SRSRAN_API int  srsran_hip_cellsearch_create(srsran_hip_cellsearch_t** h, uint32_t frame_size, uint32_t fft_size,
                                             srsran_cp_t cp, int sss_alg, uint32_t max_captures);
SRSRAN_API void srsran_hip_cellsearch_free(srsran_hip_cellsearch_t* h);
/* d_captures: n_captures x frame_size cf, contiguous.  d_cells: n_captures x 3 results (device memory).
 * n_id_2_mask: bit h set = search hypothesis h. */
SRSRAN_API int  srsran_hip_cellsearch_run(srsran_hip_cellsearch_t* h, const cf_t* d_captures, uint32_t n_captures,
                                          int n_id_2_mask, srsran_hip_cell_t* d_cells, void* stream);
/* device pointer to the correlation power of (capture, N_id_2): frame_size + fft_size - 2 floats (parity aid) */
SRSRAN_API const float* srsran_hip_cellsearch_corr(srsran_hip_cellsearch_t* h, uint32_t capture, uint32_t N_id_2);

#ifdef __cplusplus
}
#endif
#endif
