/*
 * orc_rm.c -- CPU restatement of the reference's turbo rate de-matching (receive side) and of the per-code-block
 * decode loop with CRC early stop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows lib/src/phy/fec/turbo/rm_turbo.c:175-273 (srsran_rm_turbo_gentable_receive, interleave_table_sb),
 * :390-478 (srsran_rm_turbo_rx_lut / _8bit: output[deinter[i % out_len]] += input[i], wrapping) and
 * lib/src/phy/phch/sch.c:370-492 (decode_tb_cb).  Pinned bit-exactly against oracle/_ref
 * (tests/test_oracle_golden.py).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NCOLS 32
static const uint8_t PERM[NCOLS] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
                                    1, 17, 9,  25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};

/* rm_turbo.c:175-246 with cb_len = 3K+12: position in the natural buffer [d0 d1 d2] x (K+4) of the k-th transmitted
 * soft bit of redundancy version rv_idx */
static void gentable_receive(uint16_t* table, uint32_t cb_len, uint32_t rv_idx)
{
  int nrows  = (int)((uint32_t)(cb_len / 3 - 1) / NCOLS + 1);
  int ndummy = nrows * NCOLS - (int)cb_len / 3;
  if (ndummy < 0) {
    ndummy = 0;
  }
  int       N_cb = 3 * nrows * NCOLS;
  int       k0   = nrows * (2 * (uint16_t)ceilf((float)N_cb / (float)(8 * nrows)) * rv_idx + 2);
  int       K_p  = nrows * NCOLS;
  uint16_t* t1   = malloc(sizeof(uint16_t) * 3 * 6176);
  uint16_t* t2   = malloc(sizeof(uint16_t) * 3 * 6176);
  int       k = 0, j = 0;
  while (k < (int)cb_len) {
    int jp = (k0 + j) % N_cb;
    int isdummy;
    if (jp < K_p || !(jp % 2)) {
      int d_i, d_j;
      if (jp >= K_p) {
        d_i = ((jp - K_p) / 2) / nrows;
        d_j = ((jp - K_p) / 2) % nrows;
      } else {
        d_i = jp / nrows;
        d_j = jp % nrows;
      }
      isdummy = !(d_j * NCOLS + PERM[d_i] >= ndummy);
    } else {
      uint32_t jpp  = (uint32_t)(jp - K_p - 1) / 2;
      int      kidx = (PERM[jpp / nrows] + NCOLS * (jpp % nrows) + 1) % K_p;
      isdummy       = (kidx - ndummy) < 0;
    }
    if (!isdummy) {
      t1[k] = (uint16_t)(jp % (3 * nrows * NCOLS));
      k++;
    }
    j++;
  }
  for (int i = 0; i < (int)cb_len / 3; i++) {
    int d_i = (i + ndummy) / NCOLS;
    int d_j = (i + ndummy) % NCOLS;
    for (j = 0; j < 3; j++) {
      int kidx;
      if (j != 2) {
        kidx = K_p * j + (j + 1) * (PERM[d_j] * nrows + d_i);
      } else {
        k = (i + ndummy - 1) % K_p;
        if (k < 0) {
          k += K_p;
        }
        kidx = (k / NCOLS + nrows * PERM[k % NCOLS]) % K_p;
        kidx = 2 * kidx + K_p + 1;
      }
      t2[kidx] = (uint16_t)(3 * i + j);
    }
  }
  for (int i = 0; i < (int)cb_len; i++) {
    table[i] = t2[t1[i]];
  }
  free(t1);
  free(t2);
}

int orc_rm_turbo_deinter(uint16_t* table, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb)
{
  int idx = orc_tc_cb_index(long_cb);
  if (idx < 0 || (uint32_t)orc_tc_cb_size(idx) != long_cb || rv_idx > 3 || (nof_sb && long_cb % nof_sb)) {
    return -1;
  }
  const uint32_t out_len = 3 * long_cb + 12;
  gentable_receive(table, out_len, rv_idx);
  if (nof_sb) { /* interleave_table_sb, rm_turbo.c:248-273: the layout the window decoders read directly */
    for (uint32_t i = 0; i < out_len; i++) {
      uint32_t v = table[i];
      if (v < 3 * long_cb) {
        uint32_t x = v / 3;
        table[i]   = (uint16_t)((v % 3) * (long_cb + 32) + (x % (long_cb / nof_sb)) * nof_sb + x / (long_cb / nof_sb));
      } else {
        table[i] = (uint16_t)((v - 3 * long_cb) + 3 * (long_cb + 32));
      }
    }
  }
  return 0;
}

/* srsran_rm_turbo_rx_lut_ (rm_turbo.c:405-440): nof_sb = what srsran_tdec_autoimp_get_subblocks(K) returns when the
 * decoder input layout is wanted (enable_input_tdec), 0 for the natural layout.  output is ACCUMULATED into. */
int orc_rm_turbo_rx(const int16_t* input, int16_t* output, uint32_t in_len, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb)
{
  const uint32_t out_len = 3 * long_cb + 12;
  uint16_t*      t       = malloc(sizeof(uint16_t) * out_len);
  if (orc_rm_turbo_deinter(t, long_cb, rv_idx, nof_sb)) {
    free(t);
    return -1;
  }
  for (uint32_t i = 0; i < in_len; i++) {
    output[t[i % out_len]] = (int16_t)(output[t[i % out_len]] + input[i]);
  }
  free(t);
  return 0;
}

/* srsran_rm_turbo_rx_lut_8bit (rm_turbo.c:442-478): int8, wrapping */
int orc_rm_turbo_rx_8bit(const int8_t* input, int8_t* output, uint32_t in_len, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb)
{
  const uint32_t out_len = 3 * long_cb + 12;
  uint16_t*      t       = malloc(sizeof(uint16_t) * out_len);
  if (orc_rm_turbo_deinter(t, long_cb, rv_idx, nof_sb)) {
    free(t);
    return -1;
  }
  for (uint32_t i = 0; i < in_len; i++) {
    output[t[i % out_len]] = (int8_t)(output[t[i % out_len]] + input[i]);
  }
  free(t);
  return 0;
}
