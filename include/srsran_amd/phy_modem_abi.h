/* phy_modem_abi.h -- soft demodulation and descrambling: the step between SC-FDMA de-precoding / equalisation and rate
 * de-matching.
 *
 * Reference interfaces replaced (same names, arguments, return values and arithmetic, bit for bit on an x86 build):
 *   lib/include/srsran/phy/modem/demod_soft.h:38-42    srsran_demod_soft_demodulate{,_s,_b}
 *   lib/include/srsran/phy/common/sequence.h:66-70     srsran_sequence_apply_{f,s,c}
 *   lib/include/srsran/phy/common/sequence.h:93-160    srsran_sequence_{pdsch,pusch}_apply_{f,s,c}
 * Callers in the reference: pusch.c:419-443, pdsch.c:693-744 (demodulate, then descramble in place).
 *
 * The host-pointer functions copy in, run one kernel, copy out and synchronise.  The throughput path is the batched
 * call at the end: any number of (modulation, length, seed) jobs over device-resident symbols in ONE fused pass that
 * writes the descrambled soft bits once.
 */
#ifndef SRSRAN_AMD_PHY_MODEM_ABI_H
#define SRSRAN_AMD_PHY_MODEM_ABI_H

#include "srsran_amd/phy_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* phy_common.h:285-292 */
typedef enum {
  SRSRAN_MOD_BPSK = 0,
  SRSRAN_MOD_QPSK,
  SRSRAN_MOD_16QAM,
  SRSRAN_MOD_64QAM,
  SRSRAN_MOD_256QAM,
  SRSRAN_MOD_NITEMS
} srsran_mod_t;

/* demod_soft.h:38-42.  Return 0, or -1 for an invalid modulation (demod_soft.c:846-919). */
SRSRAN_API int srsran_demod_soft_demodulate(srsran_mod_t modulation, const cf_t* symbols, float* llr, int nsymbols);
SRSRAN_API int srsran_demod_soft_demodulate_s(srsran_mod_t modulation, const cf_t* symbols, short* llr, int nsymbols);
SRSRAN_API int srsran_demod_soft_demodulate_b(srsran_mod_t modulation, const cf_t* symbols, int8_t* llr, int nsymbols);

/* sequence.h:66-70: out[i] = in[i] * (1 - 2 c(i)), c = Gold sequence of TS 36.211 7.2 with c_init = seed */
SRSRAN_API void srsran_sequence_apply_f(const float* in, float* out, uint32_t length, uint32_t seed);
SRSRAN_API void srsran_sequence_apply_s(const int16_t* in, int16_t* out, uint32_t length, uint32_t seed);
SRSRAN_API void srsran_sequence_apply_c(const int8_t* in, int8_t* out, uint32_t length, uint32_t seed);

/* sequence.h:93-160 (phch/sequences.c:63-152) */
SRSRAN_API void srsran_sequence_pdsch_apply_f(const float* in, float* out, uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id, uint32_t len);
SRSRAN_API void srsran_sequence_pdsch_apply_s(const int16_t* in, int16_t* out, uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id, uint32_t len);
SRSRAN_API void srsran_sequence_pdsch_apply_c(const int8_t* in, int8_t* out, uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id, uint32_t len);
SRSRAN_API void srsran_sequence_pusch_apply_s(const int16_t* in, int16_t* out, uint16_t rnti, uint32_t nslot, uint32_t cell_id, uint32_t len);
SRSRAN_API void srsran_sequence_pusch_apply_c(const int8_t* in, int8_t* out, uint16_t rnti, uint32_t nslot, uint32_t cell_id, uint32_t len);

/* ---- batched, device resident ---- */
#define SRSRAN_HIP_LLR_SHORT 0
#define SRSRAN_HIP_LLR_BYTE 1
#define SRSRAN_HIP_LLR_FLOAT 2
#define SRSRAN_HIP_MOD_NONE 5 /* job.mod: the input already holds soft bits of the output type (descrambling only) */
#define SRSRAN_HIP_SEQUENCE_MAX_LEN (1u << 21)

typedef struct {
  uint32_t mod;           /* srsran_mod_t, or SRSRAN_HIP_MOD_NONE */
  uint32_t nof_symbols;   /* symbols (soft bits with SRSRAN_HIP_MOD_NONE) */
  uint32_t symbol_offset; /* first symbol in d_symbols (cf_t units; soft bits with SRSRAN_HIP_MOD_NONE) */
  uint32_t llr_offset;    /* first soft bit in d_llr; multiples of 16 bytes take the fast store path */
  uint32_t seed;          /* c_init of the scrambling sequence */
  uint32_t descramble;    /* bit 0: descramble with `seed`; bit 1: change the sign of every soft bit first (the NR chain,
                           * pdsch_nr.c:467 srsran_vec_neg_bb); 0: demodulate only */
} srsran_hip_demod_job_t;

typedef struct srsran_hip_demod srsran_hip_demod_t;

SRSRAN_API int  srsran_hip_demod_create(srsran_hip_demod_t** h);
SRSRAN_API void srsran_hip_demod_free(srsran_hip_demod_t* h);
/* d_in: cf_t symbols (soft bits of llr_type for SRSRAN_HIP_MOD_NONE jobs); d_llr: soft bits of llr_type.  Asynchronous on
 * `stream`; the job list is copied before the call returns. */
SRSRAN_API int srsran_hip_demod_run(srsran_hip_demod_t* h, const void* d_in, void* d_llr, int llr_type,
                                    const srsran_hip_demod_job_t* jobs, uint32_t n_jobs, void* stream);
/* sequences.c:63-66 / 116-119 */
SRSRAN_API uint32_t srsran_hip_sequence_pdsch_seed(uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id);
SRSRAN_API uint32_t srsran_hip_sequence_pusch_seed(uint16_t rnti, uint32_t nslot, uint32_t cell_id);

/* ---- single-antenna ZF / MMSE equaliser: lib/include/srsran/phy/mimo/precoding.h:70-71 (precoding.c:357-392), the step before
 * transform de-precoding in pusch.c:413.  x = y conj(h) / ((|h|^2 + noise_estimate) scaling); csi (optional) = |h|^2 + noise.
 * Returns nof_symbols.  Float arithmetic: equals the reference to 1e-6 relative (3e-4 with csi, where the reference's SIMD body
 * uses an approximate reciprocal). */
SRSRAN_API int srsran_predecoding_single(cf_t* y, cf_t* h, cf_t* x, float* csi, int nof_symbols, float scaling, float noise_estimate);
/* device buffers, 16-byte aligned (d_csi 8-byte); asynchronous on `stream` */
SRSRAN_API int srsran_hip_predecoding_single(const cf_t* d_y, const cf_t* d_h, cf_t* d_x, float* d_csi, uint32_t nof_symbols, float scaling,
                                             float noise_estimate, void* stream);

#ifdef __cplusplus
}
#endif
#endif
