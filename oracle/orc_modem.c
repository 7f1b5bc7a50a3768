/* orc_modem.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar restatement of the soft demodulator and of the Gold-sequence descrambler:
 *   lib/src/phy/modem/demod_soft.c          (srsran_demod_soft_demodulate{,_s,_b})
 *   lib/src/phy/common/sequence.c:183-215   (sequence generator), :440-607 (apply_f / apply_s / apply_c)
 *   lib/src/phy/phch/sequences.c:63-66,116-119 (PDSCH / PUSCH seeds)
 *
 * The reference build that matters (x86, LV_HAVE_SSE) mixes two arithmetic rules inside one call: a SIMD body
 * (round-to-nearest-even conversion of symbol * -SCALE, saturating packs, integer thresholds) and a scalar tail
 * (truncating conversion of symbol * +SCALE, float thresholds for 16-QAM).  Both are restated; which one applies
 * depends only on the symbol index and nsymbols.  Out-of-range float -> integer casts (undefined in C) are modelled
 * as x86 does them: cvttss2si (0x80000000 when out of range) followed by keeping the low bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* ---- conversions ------------------------------------------------------------------------------------------------ */
static int32_t cvt_rn(float v) /* _mm_cvtps_epi32 */
{
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) {
    return INT32_MIN;
  }
  return (int32_t)rintf(v); /* default rounding mode: nearest even */
}
static int32_t cvt_tr(float v) /* _mm_cvttps_epi32 / cvttss2si */
{
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) {
    return INT32_MIN;
  }
  return (int32_t)v;
}
static int32_t cvt_tr_d(double v) /* cvttsd2si */
{
  if (!(v >= -2147483648.0 && v < 2147483648.0)) {
    return INT32_MIN;
  }
  return (int32_t)v;
}
static int16_t sat16(int32_t v) /* _mm_packs_epi32 */
{
  return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
}
static int8_t sat8(int16_t v) /* _mm_packs_epi16 */
{
  return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
}
static int16_t abs16(int16_t v) /* _mm_abs_epi16: abs(-32768) = -32768 */
{
  return (int16_t)(v < 0 ? -v : v);
}
static int8_t abs8(int8_t v)
{
  return (int8_t)(v < 0 ? -v : v);
}

#define SC_S_QPSK 100
#define SC_S_16 400
#define SC_S_64 700
#define SC_S_256 1000
#define SC_B_QPSK 20
#define SC_B_16 30
#define SC_B_64 40
#define SC_B_256 50

/* ---- int16 LLRs -------------------------------------------------------------------------------------------------- */
int orc_demod_soft_s(int mod, const float* x, int16_t* llr, int n)
{
  switch (mod) {
    case 0: /* demod_soft.c:96-101: double arithmetic, truncation */
      for (int i = 0; i < n; i++) {
        llr[i] = (int16_t)cvt_tr_d((double)(-SC_S_QPSK * (x[2 * i] + x[2 * i + 1])) * M_SQRT1_2);
      }
      return 0;
    case 1: { /* demod_soft.c:115-118 -> vector_simd.c:436-472: 16 values per AVX2 iteration saturate, the rest wraps */
      const float scale = (float)(-SC_S_QPSK * M_SQRT2);
      const int   len = 2 * n, body = len - len % 16;
      for (int i = 0; i < len; i++) {
        int32_t v = cvt_tr(x[i] * scale);
        llr[i]    = i < body ? sat16(v) : (int16_t)v;
      }
      return 0;
    }
    case 2: { /* demod_soft.c:250-299 */
      const int16_t off  = (int16_t)(2 * SC_S_16 / sqrtf(10));
      const int     body = n - n % 4;
      for (int i = 0; i < n; i++) {
        if (i < body) {
          for (int c = 0; c < 2; c++) {
            int16_t v          = sat16(cvt_rn(x[2 * i + c] * (float)-SC_S_16));
            llr[4 * i + c]     = v;
            llr[4 * i + 2 + c] = (int16_t)(abs16(v) - off);
          }
        } else {
          for (int c = 0; c < 2; c++) {
            int16_t y          = (int16_t)cvt_tr(SC_S_16 * x[2 * i + c]);
            llr[4 * i + c]     = (int16_t)-y;
            llr[4 * i + 2 + c] = (int16_t)cvt_tr((float)(y < 0 ? -(int)y : (int)y) - 2 * SC_S_16 / sqrtf(10));
          }
        }
      }
      return 0;
    }
    case 3: { /* demod_soft.c:569-644 */
      const int16_t off1 = (int16_t)(4 * SC_S_64 / sqrtf(42));
      const int16_t off2 = (int16_t)(2 * SC_S_64 / sqrtf(42));
      const int     body = n - n % 4;
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          if (i < body) {
            int16_t v          = sat16(cvt_rn(x[2 * i + c] * (float)-SC_S_64));
            int16_t a1         = (int16_t)(abs16(v) - off1);
            llr[6 * i + c]     = v;
            llr[6 * i + 2 + c] = a1;
            llr[6 * i + 4 + c] = (int16_t)(abs16(a1) - off2);
          } else {
            int16_t y          = (int16_t)cvt_tr(SC_S_64 * x[2 * i + c]);
            int16_t a1         = (int16_t)((int16_t)(y < 0 ? -(int)y : (int)y) - off1);
            llr[6 * i + c]     = (int16_t)-y;
            llr[6 * i + 2 + c] = a1;
            llr[6 * i + 4 + c] = (int16_t)((int16_t)(a1 < 0 ? -(int)a1 : (int)a1) - off2);
          }
        }
      }
      return 0;
    }
    case 4: /* demod_soft.c:824-844: float arithmetic, truncation */
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          float v            = -x[2 * i + c];
          llr[8 * i + c]     = (int16_t)cvt_tr(SC_S_256 * v);
          v                  = fabsf(v) - 8.0f / sqrtf(170.0f);
          llr[8 * i + 2 + c] = (int16_t)cvt_tr(SC_S_256 * v);
          v                  = fabsf(v) - 4.0f / sqrtf(170.0f);
          llr[8 * i + 4 + c] = (int16_t)cvt_tr(SC_S_256 * v);
          v                  = fabsf(v) - 2.0f / sqrtf(170.0f);
          llr[8 * i + 6 + c] = (int16_t)cvt_tr(SC_S_256 * v);
        }
      }
      return 0;
    default:
      return -1;
  }
}

/* ---- int8 LLRs --------------------------------------------------------------------------------------------------- */
int orc_demod_soft_b(int mod, const float* x, int8_t* llr, int n)
{
  switch (mod) {
    case 0: /* demod_soft.c:89-94 */
      for (int i = 0; i < n; i++) {
        llr[i] = (int8_t)cvt_tr_d((double)(-SC_B_QPSK * (x[2 * i] + x[2 * i + 1])) * M_SQRT1_2);
      }
      return 0;
    case 1: { /* demod_soft.c:110-113 -> vector_simd.c:524-589 (SSE, 16 values per iteration) */
      const float scale = (float)(-SC_B_QPSK * M_SQRT2);
      const int   len = 2 * n, body = len - len % 16;
      for (int i = 0; i < len; i++) {
        int32_t v = cvt_tr(x[i] * scale);
        llr[i]    = i < body ? sat8(sat16(v)) : (int8_t)v;
      }
      return 0;
    }
    case 2: { /* demod_soft.c:301-359 */
      const int8_t off  = (int8_t)(2 * SC_B_16 / sqrtf(10));
      const int    body = n - n % 8;
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          if (i < body) {
            int8_t v           = sat8(sat16(cvt_rn(x[2 * i + c] * (float)-SC_B_16)));
            llr[4 * i + c]     = v;
            llr[4 * i + 2 + c] = (int8_t)(abs8(v) - off);
          } else {
            int16_t y          = (int8_t)cvt_tr(SC_B_16 * x[2 * i + c]);
            llr[4 * i + c]     = (int8_t)-y;
            llr[4 * i + 2 + c] = (int8_t)cvt_tr((float)(y < 0 ? -y : y) - 2 * SC_B_16 / sqrtf(10));
          }
        }
      }
      return 0;
    }
    case 3: { /* demod_soft.c:646-730 */
      const int8_t off1 = (int8_t)(4 * SC_B_64 / sqrtf(42));
      const int8_t off2 = (int8_t)(2 * SC_B_64 / sqrtf(42));
      const int    body = n - n % 8;
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          if (i < body) {
            int8_t v           = sat8(sat16(cvt_rn(x[2 * i + c] * (float)-SC_B_64)));
            int8_t a1          = (int8_t)(abs8(v) - off1);
            llr[6 * i + c]     = v;
            llr[6 * i + 2 + c] = a1;
            llr[6 * i + 4 + c] = (int8_t)(abs8(a1) - off2);
          } else {
            int8_t y           = (int8_t)cvt_tr(SC_B_64 * x[2 * i + c]);
            int8_t a1          = (int8_t)((int8_t)(y < 0 ? -(int)y : (int)y) - off1);
            llr[6 * i + c]     = (int8_t)-y;
            llr[6 * i + 2 + c] = a1;
            llr[6 * i + 4 + c] = (int8_t)((int8_t)(a1 < 0 ? -(int)a1 : (int)a1) - off2);
          }
        }
      }
      return 0;
    }
    case 4: /* demod_soft.c:802-822 */
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          float v            = -x[2 * i + c];
          llr[8 * i + c]     = (int8_t)cvt_tr(SC_B_256 * v);
          v                  = fabsf(v) - 8.0f / sqrtf(170.0f);
          llr[8 * i + 2 + c] = (int8_t)cvt_tr(SC_B_256 * v);
          v                  = fabsf(v) - 4.0f / sqrtf(170.0f);
          llr[8 * i + 4 + c] = (int8_t)cvt_tr(SC_B_256 * v);
          v                  = fabsf(v) - 2.0f / sqrtf(170.0f);
          llr[8 * i + 6 + c] = (int8_t)cvt_tr(SC_B_256 * v);
        }
      }
      return 0;
    default:
      return -1;
  }
}

/* ---- float LLRs: demod_soft.c:103-108,120-136,405-417,780-800 ------------------------------------------------------ */
int orc_demod_soft_f(int mod, const float* x, float* llr, int n)
{
  switch (mod) {
    case 0:
      for (int i = 0; i < n; i++) {
        llr[i] = (float)((double)(-(x[2 * i] + x[2 * i + 1])) * M_SQRT1_2);
      }
      return 0;
    case 1: {
      const float s = (float)-M_SQRT2;
      for (int i = 0; i < 2 * n; i++) {
        llr[i] = x[i] * s;
      }
      return 0;
    }
    case 2:
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          llr[4 * i + c]     = -x[2 * i + c];
          llr[4 * i + 2 + c] = fabsf(x[2 * i + c]) - 2 / sqrtf(10);
        }
      }
      return 0;
    case 3:
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          llr[6 * i + c]     = -x[2 * i + c];
          llr[6 * i + 2 + c] = fabsf(x[2 * i + c]) - 4 / sqrtf(42);
          llr[6 * i + 4 + c] = fabsf(llr[6 * i + 2 + c]) - 2 / sqrtf(42);
        }
      }
      return 0;
    case 4:
      for (int i = 0; i < n; i++) {
        for (int c = 0; c < 2; c++) {
          float v            = -x[2 * i + c];
          llr[8 * i + c]     = v;
          v                  = fabsf(v) - 8.0f / sqrtf(170.0f);
          llr[8 * i + 2 + c] = v;
          v                  = fabsf(v) - 4.0f / sqrtf(170.0f);
          llr[8 * i + 4 + c] = v;
          v                  = fabsf(v) - 2.0f / sqrtf(170.0f);
          llr[8 * i + 6 + c] = v;
        }
      }
      return 0;
    default:
      return -1;
  }
}

/* ---- Gold sequence, TS 36.211 7.2 as sequence.c:183-215 generates it (bit-serial here) ------------------------------ */
void orc_sequence_bits(uint32_t seed, uint8_t* c, uint32_t len)
{
  /* shift registers hold x(n) .. x(n+30) in bits 0..30 */
  uint32_t x1 = 1, x2 = seed & 0x7fffffffu;
  for (uint32_t n = 0; n < 1600 + len; n++) {
    if (n >= 1600) {
      c[n - 1600] = (uint8_t)((x1 ^ x2) & 1u);
    }
    uint32_t f1 = ((x1 >> 3) ^ x1) & 1u;
    uint32_t f2 = ((x2 >> 3) ^ (x2 >> 2) ^ (x2 >> 1) ^ x2) & 1u;
    x1          = (x1 >> 1) | (f1 << 30);
    x2          = (x2 >> 1) | (f2 << 30);
  }
}

/* sequence.c:494-548 / :550-607 / :440-492: sign flip where c = 1 (two's complement wrap for -32768 / -128) */
void orc_sequence_apply_s(const int16_t* in, int16_t* out, uint32_t len, uint32_t seed, uint8_t* scratch)
{
  orc_sequence_bits(seed, scratch, len);
  for (uint32_t i = 0; i < len; i++) {
    out[i] = scratch[i] ? (int16_t)(-(int)in[i]) : in[i];
  }
}
void orc_sequence_apply_c(const int8_t* in, int8_t* out, uint32_t len, uint32_t seed, uint8_t* scratch)
{
  orc_sequence_bits(seed, scratch, len);
  for (uint32_t i = 0; i < len; i++) {
    out[i] = scratch[i] ? (int8_t)(-(int)in[i]) : in[i];
  }
}
void orc_sequence_apply_f(const float* in, float* out, uint32_t len, uint32_t seed, uint8_t* scratch)
{
  orc_sequence_bits(seed, scratch, len);
  for (uint32_t i = 0; i < len; i++) {
    uint32_t w;
    memcpy(&w, &in[i], 4);
    w ^= (uint32_t)scratch[i] << 31;
    memcpy(&out[i], &w, 4);
  }
}

/* sequences.c:63-66 and :116-119 */
uint32_t orc_sequence_pdsch_seed(uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id)
{
  return ((uint32_t)rnti << 14) + ((uint32_t)q << 13) + ((nslot / 2) << 9) + cell_id;
}
uint32_t orc_sequence_pusch_seed(uint16_t rnti, uint32_t nslot, uint32_t cell_id)
{
  return ((uint32_t)rnti << 14) + ((nslot / 2) << 9) + cell_id;
}

/* srsran_predecoding_single (mimo/precoding.c:196-392), one receive antenna: x = y conj(h) / ((|h|^2 + noise) scaling),
 * csi (optional) = |h|^2 + noise.  Evaluated in double; the reference's float paths (AVX body, scalar tail, and for the csi
 * variant an approximate reciprocal, simd.h srsran_simd_f_rcp) scatter around it by 1e-7 / 3e-4 relative. */
int orc_predecoding_single(const float* y, const float* h, float* x, float* csi, int n, float scaling, float noise_estimate)
{
  for (int i = 0; i < n; i++) {
    const double yr = y[2 * i], yi = y[2 * i + 1], hr = h[2 * i], hi = h[2 * i + 1];
    const double hh = hr * hr + hi * hi + ((csi || noise_estimate > 0) ? (double)noise_estimate : 0.0);
    const double d  = hh * (double)scaling;
    x[2 * i]        = (float)((yr * hr + yi * hi) / d);
    x[2 * i + 1]    = (float)((yi * hr - yr * hi) / d);
    if (csi) {
      csi[i] = (float)hh;
    }
  }
  return n;
}

/* ---------------- modulator (test infrastructure) ----------------
 * Constellation tables of lib/src/phy/modem/lte_tables.c:30-181 (TS 36.211 7.1.1-7.1.5), restated from the standard's closed forms
 *   16-QAM  I = (1 - 2 b0)(2 - (1 - 2 b2)) / sqrt 10                 Q likewise with b1, b3
 *   64-QAM  I = (1 - 2 b0)(4 - (1 - 2 b2)(2 - (1 - 2 b4))) / sqrt 42
 *   256-QAM I = (1 - 2 b0)(8 - (1 - 2 b2)(4 - (1 - 2 b4)(2 - (1 - 2 b6)))) / sqrt 170
 * with the levels evaluated as the reference does: float k / sqrtf(N) (lte_tables.h:30-36), (float)M_SQRT1_2 for BPSK / QPSK.
 * table index = the symbol's bits, first bit most significant (modem_table.c:60-140 fills symbol_table that way).  out: 2^Qm (re, im). */
static int orc_qm(int mod)
{
  return mod == 0 ? 1 : 2 * mod;
}

int orc_mod_table(int mod, float* out)
{
  if (mod < 0 || mod > 4) {
    return -1;
  }
  const int qm = orc_qm(mod), n = 1 << qm;
  for (int i = 0; i < n; i++) {
    int b[8];
    for (int k = 0; k < qm; k++) {
      b[k] = (i >> (qm - 1 - k)) & 1;
    }
    float re, im;
    if (mod == 0) {
      re = im = b[0] ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
    } else if (mod == 1) {
      re = b[0] ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
      im = b[1] ? -(float)M_SQRT1_2 : (float)M_SQRT1_2;
    } else {
      /* amplitude in units of 1 / sqrt N from the nested form, as an odd integer; the level itself is k / sqrtf(N) in float */
      int   a[2];
      float norm = mod == 2 ? sqrtf(10.0f) : (mod == 3 ? sqrtf(42.0f) : sqrtf(170.0f));
      for (int c = 0; c < 2; c++) {
        int v = 1; /* innermost (2 - (1 - 2 b)) starts from the last pair */
        v     = 2 - (1 - 2 * b[qm - 2 + c]);
        for (int k = qm / 2 - 2; k >= 1; k--) {
          v = (1 << (qm / 2 - k)) - (1 - 2 * b[2 * k + c]) * v;
        }
        a[c] = v;
      }
      re = (float)a[0] / norm;
      im = (float)a[1] / norm;
      if (b[0]) {
        re = -re;
      }
      if (b[1]) {
        im = -im;
      }
    }
    out[2 * i]     = re;
    out[2 * i + 1] = im;
  }
  return n;
}

/* srsran_sequence_apply_packed (sequence.c) + srsran_mod_modulate_bytes (mod.c:135-166) + srsran_vec_sc_prod_cfc: byte-packed bits, MSB first */
int orc_modulate_bytes(int mod, const uint8_t* bits, float* out, uint32_t nbits, uint32_t seed, int scramble, float scaling)
{
  float tab[512];
  if (orc_mod_table(mod, tab) < 0 || nbits % (uint32_t)orc_qm(mod)) {
    return -1;
  }
  const uint32_t qm = (uint32_t)orc_qm(mod), n = nbits / qm;
  uint8_t*       c = (uint8_t*)calloc(nbits + 1, 1);
  if (scramble) {
    orc_sequence_bits(seed, c, nbits);
  }
  for (uint32_t s = 0; s < n; s++) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < qm; k++) {
      const uint32_t b = s * qm + k;
      v                = (v << 1) | (((bits[b >> 3] >> (7 - (b & 7))) & 1u) ^ c[b]);
    }
    out[2 * s]     = scaling == 1.0f ? tab[2 * v] : tab[2 * v] * scaling;
    out[2 * s + 1] = scaling == 1.0f ? tab[2 * v + 1] : tab[2 * v + 1] * scaling;
  }
  free(c);
  return (int)n;
}

/* UL-SCH channel interleaver, TS 36.212 5.2.2.8 without RI / ACK bits (sch.c:660-681 ulsch_interleave_gen with ri_present = NULL): position of
 * every bit of q in g.  nof_sym groups of Qm bits; the matrix has `cols` columns (SC-FDMA symbols) and is written row by row, read column by column. */
int orc_ulsch_interleaver_lut(uint32_t nof_sym, uint32_t Qm, uint32_t cols, uint32_t* lut)
{
  if (cols == 0 || nof_sym % cols) {
    return -1;
  }
  const uint32_t rows = nof_sym / cols;
  uint32_t       idx  = 0;
  for (uint32_t j = 0; j < rows; j++) {
    for (uint32_t i = 0; i < cols; i++) {
      for (uint32_t k = 0; k < Qm; k++) {
        lut[(i * rows + j) * Qm + k] = idx++;
      }
    }
  }
  return 0;
}
