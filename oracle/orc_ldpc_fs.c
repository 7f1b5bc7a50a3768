/*
 * orc_ldpc_fs.c -- CPU restatement of the reference's float and int16 layered LDPC decoders.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows lib/src/phy/fec/ldpc/ldpc_dec_f.c and ldpc_dec_s.c under the schedule of ldpc_decoder.c:44-104
 * (srsran_ldpc_decoder_decode_f / _decode_s, no CRC).  Same structure as the int8 decoder (orc_ldpc.c); what
 * changes per type is the variable-to-check clipping, the scaling and the soft-bit saturation.
 * Pinned bit-exactly against oracle/_ref (tests/test_oracle_golden.py); the float decoder only uses IEEE
 * subtract / multiply / add, so bit-exactness is well defined for it too.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DEFINE_DECODER(NAME, T, MIN_INIT, ABS, V2C, SCALE, SOFT)                                                       \
  int NAME(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const T* llrs, uint8_t* message,          \
           uint32_t cdwd_rm_length, T* soft_out)                                                                       \
  {                                                                                                                    \
    const int ls = g->ls, bgN = g->bgN, bgM = g->bgM, bgK = g->bgK;                                                    \
    const int liftN = bgN * ls, liftK = bgK * ls, hrrN = (bgK + 4) * ls;                                               \
    if (max_nof_iter == 0) {                                                                                           \
      max_nof_iter = 10;                                                                                               \
    }                                                                                                                  \
    const int   sf_i = (int)(scaling_fctr * 100); /* ldpc_dec_s.c:150 */                                              \
    const float sf_f = scaling_fctr;                                                                                   \
    (void)sf_i;                                                                                                        \
    (void)sf_f;                                                                                                        \
    if (cdwd_rm_length > (uint32_t)(liftN - 2 * ls)) {                                                                 \
      cdwd_rm_length = liftN - 2 * ls;                                                                                 \
    }                                                                                                                  \
    if (cdwd_rm_length < (uint32_t)((bgK + 2) * ls)) {                                                                 \
      cdwd_rm_length = (bgK + 2) * ls;                                                                                 \
    }                                                                                                                  \
    if (cdwd_rm_length % ls) {                                                                                         \
      cdwd_rm_length = (cdwd_rm_length / ls + 1) * ls;                                                                 \
    }                                                                                                                  \
    const int n_layers = (uint8_t)(cdwd_rm_length / ls - bgK + 2);                                                     \
    T*   soft    = malloc(sizeof(T) * liftN);                                                                          \
    T*   c2v     = calloc((size_t)(hrrN + ls) * bgM, sizeof(T));                                                       \
    T*   v2c     = calloc(hrrN + ls, sizeof(T));                                                                       \
    T(*minv)[2]  = malloc(ls * sizeof(T[2]));                                                                          \
    int* min_idx = calloc(ls, sizeof(int));                                                                            \
    int* prod    = malloc(ls * sizeof(int));                                                                           \
    memset(soft, 0, sizeof(T) * 2 * ls);                                                                               \
    memcpy(soft + 2 * ls, llrs, sizeof(T) * (liftN - 2 * ls));                                                         \
    for (int it = 0; it < max_nof_iter; it++) {                                                                        \
      for (int l = 0; l < n_layers; l++) {                                                                             \
        T* this_c2v = c2v + (size_t)l * (hrrN + ls);                                                                   \
        for (int i = 0; i < hrrN + (l >= 4 ? ls : 0); i++) {                                                           \
          T x    = (i < hrrN) ? soft[i] : soft[hrrN + (l - 4) * ls + (i - hrrN)];                                      \
          T y    = this_c2v[i];                                                                                        \
          v2c[i] = V2C(x, y);                                                                                          \
        }                                                                                                              \
        for (int i = 0; i < ls; i++) {                                                                                 \
          prod[i]    = 1;                                                                                              \
          minv[i][0] = minv[i][1] = MIN_INIT;                                                                          \
        }                                                                                                              \
        for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {                                                  \
          int shift = g->shift[e];                                                                                     \
          int base  = g->col[e] * ls;                                                                                  \
          base      = base <= hrrN ? base : hrrN;                                                                      \
          for (int j = 0; j < ls; j++) {                                                                               \
            int index      = (j + ls - shift) % ls;                                                                    \
            int iv         = base + j;                                                                                 \
            T   a          = (T)ABS(v2c[iv]);                                                                          \
            int is_min     = a < minv[index][0];                                                                       \
            minv[index][1] = (a >= minv[index][1]) ? minv[index][1] : (is_min ? minv[index][0] : a);                   \
            minv[index][0] = is_min ? a : minv[index][0];                                                              \
            min_idx[index] = is_min ? iv : min_idx[index];                                                             \
            prod[index] *= (v2c[iv] >= 0) ? 1 : -1;                                                                    \
          }                                                                                                            \
        }                                                                                                              \
        for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {                                                  \
          int shift = g->shift[e];                                                                                     \
          int base  = g->col[e] * ls;                                                                                  \
          base      = base <= hrrN ? base : hrrN;                                                                      \
          for (int j = 0; j < ls; j++) {                                                                               \
            int index    = (j + ls - shift) % ls;                                                                      \
            int iv       = base + j;                                                                                   \
            T   m        = (iv != min_idx[index]) ? minv[index][0] : minv[index][1];                                   \
            m            = SCALE(m);                                                                                   \
            this_c2v[iv] = (T)(m * (T)(prod[index] * ((v2c[iv] >= 0) ? 1 : -1)));                                      \
          }                                                                                                            \
        }                                                                                                              \
        for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {                                                  \
          int ext = g->col[e] * ls;                                                                                    \
          for (int j = 0; j < ls; j++) {                                                                               \
            int ib   = ext + j;                                                                                        \
            int it2  = (ext <= hrrN) ? ib : hrrN + j;                                                                  \
            soft[ib] = SOFT(this_c2v[it2], v2c[it2]);                                                                  \
          }                                                                                                            \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    for (int i = 0; i < liftK; i++) {                                                                                  \
      message[i] = (soft[i] < 0);                                                                                      \
    }                                                                                                                  \
    if (soft_out) {                                                                                                    \
      memcpy(soft_out, soft, sizeof(T) * liftN);                                                                       \
    }                                                                                                                  \
    free(soft);                                                                                                        \
    free(c2v);                                                                                                         \
    free(v2c);                                                                                                         \
    free(minv);                                                                                                        \
    free(min_idx);                                                                                                     \
    free(prod);                                                                                                        \
    return max_nof_iter;                                                                                               \
  }

/* ---- int16: ldpc_dec_s.c.  Messages 15 bit (|.| <= 16383), soft bits use +-32767 as infinity */
static inline int16_t v2c_s(int16_t x, int16_t y)
{
  if (x >= 32767) {
    return 32767; /* inner_var_to_check_s, :338-363 */
  }
  if (x <= -32767) {
    return -32767;
  }
  long t = (long)x - y;
  return (int16_t)(t > 16383 ? 16383 : (t < -16383 ? -16383 : t));
}
static inline int16_t soft_s(int16_t c, int16_t v)
{
  long t = (long)c + v; /* update_ldpc_soft_bits_s :286-321 */
  if (t > 16383) {
    t = INT16_MAX;
  }
  if (t < -16383) {
    t = -INT16_MAX;
  }
  return (int16_t)t;
}
#define SCALE_S(m) ((int16_t)((m)*sf_i / 100)) /* :276: this_check_to_var * scaling_fctr / F2I */
DEFINE_DECODER(orc_ldpc_decode_s, int16_t, INT16_MAX, abs, v2c_s, SCALE_S, soft_s)

/* ---- float: ldpc_dec_f.c.  No clipping anywhere */
static inline float v2c_f(float x, float y)
{
  return x - y; /* srsran_vec_sub_fff, :172-181 */
}
static inline float soft_f(float c, float v)
{
  return c + v; /* :278 */
}
#define SCALE_F(m) ((m)*sf_f) /* :246 */
DEFINE_DECODER(orc_ldpc_decode_f, float, INFINITY, fabsf, v2c_f, SCALE_F, soft_f)
