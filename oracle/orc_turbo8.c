/*
 * orc_turbo8.c -- CPU restatement of the reference's 8-bit LLR turbo decoders.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows lib/include/srsran/phy/fec/turbo/turbodecoder_win.h with WINIMP_IS_SSE8 (16 sub-blocks, :154-193)
 * and WINIMP_IS_AVX8 (32 sub-blocks, :196-300): saturating int8 adds, INF = 0, normalisation by the
 * maximum state metric at every step but the first, LLR output halved (divide_output 1); and
 * turbodecoder_iter.h:72-141 with LLR_IS_8BIT (srsran_vec_sub_bbb / srsran_vec_lut_bbb) plus the dispatch of
 * turbodecoder.c:410-512 (8-bit API falls back to the 16-bit decoders for K the 8-bit ones do not take).
 * Pinned bit-exactly against oracle/_ref (tests/test_oracle_golden.py).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define WIN_OVERLAP 40

static inline int8_t adds8(int a, int b)
{
  int z = a + b;
  return (int8_t)(z > 127 ? 127 : (z < -128 ? -128 : z));
}
static inline int8_t subs8(int a, int b)
{
  int z = a - b;
  return (int8_t)(z > 127 ? 127 : (z < -128 ? -128 : z));
}
static inline int8_t max8(int8_t a, int8_t b)
{
  return a > b ? a : b;
}
/* turbodecoder_win.h:470-478: the scalar helper only clamps the positive side */
static inline int8_t sadd8(int8_t x, int8_t y)
{
  int16_t z = (int16_t)((int16_t)x + y);
  return z > 127 ? 127 : (int8_t)z;
}

/* turbodecoder_win.h:480-498 with normalize_max, normalize_period 1 */
static void normalize8(int nb, uint32_t k, int8_t old[8][32])
{
  if (k == 0) {
    return;
  }
  for (int d = 0; d < nb; d++) {
    int8_t m = max8(old[0][d], old[1][d]);
    for (int i = 2; i < 8; i++) {
      m = max8(m, old[i][d]);
    }
    for (int i = 0; i < 8; i++) {
      old[i][d] = subs8(old[i][d], m);
    }
  }
}

/* turbodecoder_win.h:500-548 (INF = 0: every state starts at 0) */
static void beta_trellis8(const int8_t* input, const int8_t* parity, uint32_t K, int8_t old[8])
{
  int8_t m_b[8], nw[8];
  for (int i = 0; i < 8; i++) {
    old[i] = 0;
  }
  for (int k = (int)K + 2; k >= (int)K; k--) {
    int8_t x = input[k], y = parity[k];
    int8_t xy = sadd8(x, y);
    m_b[0] = sadd8(old[4], xy);
    m_b[1] = old[4];
    m_b[2] = sadd8(old[5], y);
    m_b[3] = sadd8(old[5], x);
    m_b[4] = sadd8(old[6], x);
    m_b[5] = sadd8(old[6], y);
    m_b[6] = old[7];
    m_b[7] = sadd8(old[7], xy);
    nw[0] = old[0];
    nw[1] = sadd8(old[0], xy);
    nw[2] = sadd8(old[1], x);
    nw[3] = sadd8(old[1], y);
    nw[4] = sadd8(old[2], y);
    nw[5] = sadd8(old[2], x);
    nw[6] = sadd8(old[3], xy);
    nw[7] = old[3];
    for (int i = 0; i < 8; i++) {
      old[i] = m_b[i] > nw[i] ? m_b[i] : nw[i];
    }
  }
}

/* one backward step for all sub-blocks, turbodecoder_win.h:626-652 */
static void beta_step8(int nb, const int8_t* in_k, const int8_t* app_k, const int8_t* par_k, int8_t old[8][32])
{
  for (int d = 0; d < nb; d++) {
    int8_t x = in_k[d], y = par_k[d];
    if (app_k) {
      x = adds8(app_k[d], x);
    }
    int8_t xy = adds8(x, y);
    int8_t o[8], m_b[8], nw[8];
    for (int i = 0; i < 8; i++) {
      o[i] = old[i][d];
    }
    m_b[0] = adds8(o[4], xy);
    m_b[1] = o[4];
    m_b[2] = adds8(o[5], y);
    m_b[3] = adds8(o[5], x);
    m_b[4] = adds8(o[6], x);
    m_b[5] = adds8(o[6], y);
    m_b[6] = o[7];
    m_b[7] = adds8(o[7], xy);
    nw[0] = o[0];
    nw[1] = adds8(o[0], xy);
    nw[2] = adds8(o[1], x);
    nw[3] = adds8(o[1], y);
    nw[4] = adds8(o[2], y);
    nw[5] = adds8(o[2], x);
    nw[6] = adds8(o[3], xy);
    nw[7] = o[3];
    for (int i = 0; i < 8; i++) {
      old[i][d] = max8(m_b[i], nw[i]);
    }
  }
}

/* turbodecoder_win.h:551-681.  beta: [(long_sb+1)][8][nb] */
static void win_beta8(int nb, const int8_t* input, const int8_t* app, const int8_t* parity, uint32_t K, int8_t* beta)
{
  uint32_t long_sb = K / nb;
  int8_t   old[8][32];
  memset(old, 0, sizeof(old)); /* -INF = 0 */
  for (int k = WIN_OVERLAP - 1; k >= 0; k--) {
    beta_step8(nb, &input[nb * k], app ? &app[nb * k] : NULL, &parity[nb * k], old);
    normalize8(nb, (uint32_t)k, old);
  }
  int8_t tr[8];
  beta_trellis8(input, parity, K, tr);
  for (int i = 0; i < 8; i++) {
    for (int d = 0; d < nb - 1; d++) {
      old[i][d] = old[i][d + 1]; /* the 128-bit lane crossing is patched by hand in the reference (:590-617) */
    }
    old[i][nb - 1] = tr[i];
    memcpy(&beta[(8 * long_sb + i) * nb], old[i], nb);
  }
  for (int k = (int)long_sb - 1; k >= 0; k--) {
    beta_step8(nb, &input[nb * k], app ? &app[nb * k] : NULL, &parity[nb * k], old);
    for (int i = 0; i < 8; i++) {
      memcpy(&beta[(8 * k + i) * nb], old[i], nb);
    }
    normalize8(nb, (uint32_t)k, old);
  }
}

/* turbodecoder_win.h:684-832 */
static void win_alpha8(int nb, const int8_t* input, const int8_t* app, const int8_t* parity, int8_t* output, uint32_t K,
                       const int8_t* beta)
{
  uint32_t long_sb = K / nb;
  int8_t   old[8][32];
  for (int pass = 0; pass < 2; pass++) {
    uint32_t loop_len = pass ? long_sb : WIN_OVERLAP;
    if (pass) {
      for (int i = 0; i < 8; i++) {
        for (int d = nb - 1; d > 0; d--) {
          old[i][d] = old[i][d - 1];
        }
        old[i][0] = 0; /* state 0 known = 0, the others -INF = 0 */
      }
    } else {
      memset(old, 0, sizeof(old));
    }
    uint32_t base = long_sb - loop_len;
    for (uint32_t k = 0; k < loop_len; k++) {
      const int8_t* in_k  = &input[nb * (base + k)];
      const int8_t* par_k = &parity[nb * (base + k)];
      const int8_t* app_k = app ? &app[nb * (base + k)] : NULL;
      for (int d = 0; d < nb; d++) {
        int8_t x = in_k[d], y = par_k[d];
        if (app_k) {
          x = adds8(app_k[d], x);
        }
        int8_t xy = adds8(x, y);
        int8_t o[8], m_b[8], nw[8];
        for (int i = 0; i < 8; i++) {
          o[i] = old[i][d];
        }
        m_b[0] = o[0];
        m_b[1] = adds8(o[3], y);
        m_b[2] = adds8(o[4], y);
        m_b[3] = o[7];
        m_b[4] = o[1];
        m_b[5] = adds8(o[2], y);
        m_b[6] = adds8(o[5], y);
        m_b[7] = o[6];
        nw[0] = adds8(o[1], xy);
        nw[1] = adds8(o[2], x);
        nw[2] = adds8(o[5], x);
        nw[3] = adds8(o[6], xy);
        nw[4] = adds8(o[0], xy);
        nw[5] = adds8(o[3], x);
        nw[6] = adds8(o[4], x);
        nw[7] = adds8(o[7], xy);
        if (pass) {
          int8_t m1 = 0, m0 = 0;
          for (int i = 0; i < 8; i++) {
            int8_t b  = beta[(8 * (k + 1) + i) * nb + d];
            int8_t v0 = adds8(b, m_b[i]);
            int8_t v1 = adds8(b, nw[i]);
            m0 = (i == 0) ? v0 : max8(m0, v0);
            m1 = (i == 0) ? v1 : max8(m1, v1);
          }
          output[nb * k + d] = (int8_t)(subs8(m1, m0) >> 1); /* simd_rb_shift(out, 1), :180-184,812 */
        }
        for (int i = 0; i < 8; i++) {
          old[i][d] = max8(m_b[i], nw[i]);
        }
      }
      normalize8(nb, k, old);
    }
  }
}

/* srsran_vec_sub_bbb on an AVX2 host (vector_simd.c:162-190): saturating in whole 32-byte vectors, the
 * remainder wraps */
static void vec_sub_bbb(const int8_t* x, const int8_t* y, int8_t* z, uint32_t len)
{
  uint32_t i = 0, nv = (len / 32) * 32;
  for (; i < nv; i++) {
    z[i] = subs8(x[i], y[i]);
  }
  for (; i < len; i++) {
    z[i] = (int8_t)(x[i] - y[i]);
  }
}

int orc_tdec_run_all_8bit(const int8_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t K, int impl,
                          int sb_layout, int16_t* dec_llr)
{
  int cbidx = orc_tc_cb_index(K);
  if (cbidx < 0 || K > 6144 || (uint32_t)orc_tc_cb_size(cbidx) != K) {
    return -1;
  }
  uint32_t nb;
  switch (impl) {
    case ORC_TDEC_AUTO:
      nb = orc_tdec_autoimp_subblocks_8bit(K);
      if (nb < 16) {
        /* turbodecoder.c:455-478: no 8-bit decoder takes this K -> widen and run the 16-bit one */
        uint32_t len = (sb_layout ? 3 * (K + 32) : 3 * K) + 12;
        int16_t* in16 = malloc(len * 2);
        for (uint32_t i = 0; i < len; i++) {
          in16[i] = input[i];
        }
        int rc = orc_tdec_run_all(in16, output, nof_iterations, K, nb == 8 ? ORC_TDEC_SSE_WINDOW : ORC_TDEC_GENERIC,
                                  sb_layout, NULL, dec_llr);
        free(in16);
        return rc;
      }
      break;
    case ORC_TDEC_SSE8_WINDOW:
      nb = 16;
      break;
    case ORC_TDEC_AVX8_WINDOW:
      nb = 32;
      break;
    default:
      return -1;
  }
  if (K % nb || K / nb <= WIN_OVERLAP) { /* == 40: the reference takes its main-pass branch twice (:573,:712) */
    return -1;
  }
  uint32_t  len  = K + 12;
  int8_t*   syst = calloc(len, 1), *par0 = calloc(len, 1), *par1 = calloc(len, 1);
  int8_t*   app1 = calloc(len, 1), *app2 = calloc(len, 1), *ext1 = calloc(len, 1), *ext2 = calloc(len, 1);
  uint16_t* inter = malloc(K * 2), *deinter = malloc(K * 2);
  int8_t*   beta = malloc(8 * (K + 64));
  orc_qpp_gen(K, nb, inter, deinter);
  uint32_t long_sb = K / nb;

  if (sb_layout) { /* turbodecoder_iter.h:58-70,88-102 */
    memcpy(syst, input, K);
    memcpy(par0, &input[K + 32], K);
    memcpy(par1, &input[2 * (K + 32)], K);
  } else { /* turbodecoder_win.h:880-930 */
    for (uint32_t n = 0; n < K; n++) {
      uint32_t idx = (n % long_sb) * nb + n / long_sb;
      syst[idx] = input[3 * n];
      par0[idx] = input[3 * n + 1];
      par1[idx] = input[3 * n + 2];
    }
  }
  uint32_t tb = sb_layout ? 3 * (K + 32) : 3 * K;
  for (uint32_t i = K; i < K + 3; i++) {
    syst[i] = input[tb + 2 * (i - K)];
    par0[i] = input[tb + 2 * (i - K) + 1];
    app2[i] = input[tb + 6 + 2 * (i - K)];
    par1[i] = input[tb + 6 + 2 * (i - K) + 1];
  }

  uint32_t n_iter = 0;
  do {
    if ((n_iter % 2) == 0) {
      if (n_iter) {
        vec_sub_bbb(app1, ext1, app1, K);
      }
      win_beta8(nb, syst, n_iter ? app1 : NULL, par0, K, beta);
      win_alpha8(nb, syst, n_iter ? app1 : NULL, par0, ext1, K, beta);
    } else {
      if (n_iter > 1) {
        vec_sub_bbb(ext1, app1, ext1, K);
      }
      for (uint32_t i = 0; i < K; i++) {
        app2[deinter[i]] = ext1[i];
      }
      win_beta8(nb, app2, NULL, par1, K, beta);
      win_alpha8(nb, app2, NULL, par1, ext2, K, beta);
      for (uint32_t i = 0; i < K; i++) {
        app1[inter[i]] = ext2[i];
      }
    }
    n_iter++;
  } while (n_iter < nof_iterations);

  const int8_t* dec = !(n_iter % 2) ? app1 : ext1;
  memset(output, 0, K / 8);
  for (uint32_t n = 0; n < K; n++) {
    uint32_t idx = (n % long_sb) * nb + n / long_sb;
    if (dec[idx] > 0) {
      output[n / 8] |= (uint8_t)(0x80u >> (n % 8));
    }
    if (dec_llr) {
      dec_llr[n] = dec[idx];
    }
  }
  free(syst);
  free(par0);
  free(par1);
  free(app1);
  free(app2);
  free(ext1);
  free(ext2);
  free(inter);
  free(deinter);
  free(beta);
  return 0;
}
