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

/* ------------------------------------------------------------------ transport block (sch.c:370-560) */

/* srsran_cbsegm, cbsegm.c:62-117 */
int orc_cbsegm(uint32_t tbs, uint32_t* C, uint32_t* K1, uint32_t* K2, uint32_t* C1, uint32_t* C2, uint32_t* F)
{
  *C = *K1 = *K2 = *C1 = *C2 = *F = 0;
  if (tbs == 0) {
    return 0;
  }
  uint32_t B = tbs + 24, Bp;
  if (B <= 6144) {
    *C = 1;
    Bp = B;
  } else {
    *C = (B + (6144 - 24) - 1) / (6144 - 24);
    Bp = B + 24 * (*C);
  }
  int idx1 = orc_tc_cb_index((Bp - 1) / (*C) + 1);
  if (idx1 < 0) {
    return -1;
  }
  *K1 = (uint32_t)orc_tc_cb_size(idx1);
  if (*C == 1) {
    *C1 = 1;
  } else {
    *K2 = (uint32_t)orc_tc_cb_size(idx1 > 0 ? idx1 - 1 : idx1);
    *C2 = (*K1 != *K2) ? ((*C) * (*K1) - Bp) / (*K1 - *K2) : 0;
    *C1 = *C - *C2;
  }
  *F = (*C1) * (*K1) + (*C2) * (*K2) - Bp;
  return 0;
}

static uint32_t crc_bytes(uint32_t poly, const uint8_t* data, uint32_t nbits)
{
  uint8_t* bits = malloc(nbits);
  for (uint32_t i = 0; i < nbits; i++) {
    bits[i] = (data[i / 8] >> (7 - (i % 8))) & 1;
  }
  uint32_t c = orc_crc_bits(poly, 24, bits, (int)nbits); /* srsran_crc_checksum_byte, crc.c:147-161 */
  free(bits);
  return c;
}

/* decode_tb + decode_tb_cb for one transport block, 16-bit LLRs, window decoders (K > 400).
 *   softbuf : C slots of 18600 int16 (accumulated into), cb_crc: C flags (read and updated)
 *   cb_data : C x 768 bytes, the decoded code blocks kept between HARQ rounds (softbuffer->data)
 *   data    : tbs/8 + 6 bytes
 * returns 0 (SRSRAN_SUCCESS) or -1; *avg_iterations as sch.c:485 */
int orc_sch_decode_tb(uint32_t tbs, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, const int16_t* e_bits, int16_t* softbuf,
                      uint8_t* cb_crc, uint8_t* cb_data, uint32_t max_iterations, uint8_t* data, float* avg_iterations)
{
  uint32_t C, K1, K2, C1, C2, F;
  if (orc_cbsegm(tbs, &C, &K1, &K2, &C1, &C2, &F) || F || Qm == 0) {
    return -2;
  }
  float it = 0;
  for (uint32_t i = 0; i < C; i++) {
    uint32_t K    = i < C1 ? K1 : K2;
    uint32_t rlen = C == 1 ? K : K - 24;
    if (cb_crc[i]) {
      memcpy(&data[i * rlen / 8], &cb_data[(size_t)i * 768], rlen / 8); /* sch.c:466-471 */
      continue;
    }
    uint32_t Gp = nof_e_bits / Qm, gamma = Gp % C, n_e = Qm * (Gp / C);
    uint32_t rp = i * n_e, n_e2 = n_e;
    if (i > C - gamma) {
      n_e2 = n_e + Qm;
      rp   = (C - gamma) * n_e + (i - (C - gamma)) * n_e2;
    }
    int16_t* sb  = softbuf + (size_t)i * 18600;
    uint32_t nsb = orc_tdec_autoimp_subblocks(K);
    /* nsb == 0 (K <= 400): natural soft-buffer layout (rm_turbo.c:412-421) and the scalar decoder (turbodecoder.c:381-408) */
    if (orc_rm_turbo_rx(&e_bits[rp], sb, n_e2, K, rv, nsb)) {
      return -2;
    }
    uint8_t* out = &data[i * rlen / 8];
    uint32_t noi = 0;
    int      ok  = 0;
    do {
      noi++;
      /* srsran_tdec_iteration noi times from the start = the state after noi half iterations */
      if (orc_tdec_run_all(sb, out, noi, K, ORC_TDEC_AUTO, nsb ? 1 : 0, NULL, NULL)) {
        return -2;
      }
      it += 1;
      ok = crc_bytes(C > 1 ? 0x1800063 : 0x1864CFB, out, C > 1 ? K : tbs + 24) == 0;
    } while (noi < max_iterations && !ok);
    if (ok) {
      cb_crc[i] = 1;
    }
  }
  *avg_iterations = it / (float)C;
  int all = 1;
  for (uint32_t i = 0; i < C; i++) {
    all = all && cb_crc[i];
  }
  if (!all) { /* sch.c:478-485: keep the good code blocks for the next HARQ round */
    for (uint32_t i = 0; i < C; i++) {
      if (cb_crc[i]) {
        uint32_t K = i < C1 ? K1 : K2, rlen = C == 1 ? K : K - 24;
        memcpy(&cb_data[(size_t)i * 768], &data[i * rlen / 8], rlen / 8);
      }
    }
    return -1;
  }
  uint32_t par_rx = crc_bytes(0x1864CFB, data, tbs);
  uint32_t par_tx = ((uint32_t)data[tbs / 8] << 16) | ((uint32_t)data[tbs / 8 + 1] << 8) | data[tbs / 8 + 2];
  return (par_rx == par_tx && par_rx) ? 0 : -1;
}

/* the same loop with q->llr_is_8bit (sch.c:408-412,426-428): 8-bit rate de-matching and the 8-bit window decoders */
int orc_sch_decode_tb_8bit(uint32_t tbs, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, const int8_t* e_bits, int8_t* softbuf,
                      uint8_t* cb_crc, uint8_t* cb_data, uint32_t max_iterations, uint8_t* data, float* avg_iterations)
{
  uint32_t C, K1, K2, C1, C2, F;
  if (orc_cbsegm(tbs, &C, &K1, &K2, &C1, &C2, &F) || F || Qm == 0) {
    return -2;
  }
  float it = 0;
  for (uint32_t i = 0; i < C; i++) {
    uint32_t K    = i < C1 ? K1 : K2;
    uint32_t rlen = C == 1 ? K : K - 24;
    if (cb_crc[i]) {
      memcpy(&data[i * rlen / 8], &cb_data[(size_t)i * 768], rlen / 8); /* sch.c:466-471 */
      continue;
    }
    uint32_t Gp = nof_e_bits / Qm, gamma = Gp % C, n_e = Qm * (Gp / C);
    uint32_t rp = i * n_e, n_e2 = n_e;
    if (i > C - gamma) {
      n_e2 = n_e + Qm;
      rp   = (C - gamma) * n_e + (i - (C - gamma)) * n_e2;
    }
    int8_t*  sb  = softbuf + (size_t)i * 18600;
    uint32_t nsb = orc_tdec_autoimp_subblocks_8bit(K);
    /* nsb == 0 (K <= 400): natural layout; the 8-bit API widens to int16 and runs the scalar decoder (turbodecoder.c:455-478) */
    if (orc_rm_turbo_rx_8bit(&e_bits[rp], sb, n_e2, K, rv, nsb)) {
      return -2;
    }
    uint8_t* out = &data[i * rlen / 8];
    uint32_t noi = 0;
    int      ok  = 0;
    do {
      noi++;
      /* srsran_tdec_iteration noi times from the start = the state after noi half iterations */
      if (orc_tdec_run_all_8bit(sb, out, noi, K, ORC_TDEC_AUTO, nsb ? 1 : 0, NULL)) {
        return -2;
      }
      it += 1;
      ok = crc_bytes(C > 1 ? 0x1800063 : 0x1864CFB, out, C > 1 ? K : tbs + 24) == 0;
    } while (noi < max_iterations && !ok);
    if (ok) {
      cb_crc[i] = 1;
    }
  }
  *avg_iterations = it / (float)C;
  int all = 1;
  for (uint32_t i = 0; i < C; i++) {
    all = all && cb_crc[i];
  }
  if (!all) { /* sch.c:478-485: keep the good code blocks for the next HARQ round */
    for (uint32_t i = 0; i < C; i++) {
      if (cb_crc[i]) {
        uint32_t K = i < C1 ? K1 : K2, rlen = C == 1 ? K : K - 24;
        memcpy(&cb_data[(size_t)i * 768], &data[i * rlen / 8], rlen / 8);
      }
    }
    return -1;
  }
  uint32_t par_rx = crc_bytes(0x1864CFB, data, tbs);
  uint32_t par_tx = ((uint32_t)data[tbs / 8] << 16) | ((uint32_t)data[tbs / 8 + 1] << 8) | data[tbs / 8 + 2];
  return (par_rx == par_tx && par_rx) ? 0 : -1;
}
