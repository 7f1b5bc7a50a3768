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
