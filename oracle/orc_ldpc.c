/*
 * orc_ldpc.c -- scalar restatement of the reference NR LDPC int8 layered min-sum decoder
 * (TEST INFRASTRUCTURE ONLY).
 *
 * Restates:
 *   lib/src/phy/fec/ldpc/ldpc_decoder.c:44-104   (schedule: clamp of cdwd_rm_length, n_layers, CRC early stop)
 *   lib/src/phy/fec/ldpc/ldpc_dec_c.c:170-363    (init, var->check, check->var, soft bits, message)
 *   lib/src/phy/fec/ldpc/base_graph.c:4467-4503  (compact PCM = V mod Z) -- constants from
 *                                                 srslte_amd/csrc/tables/nr_ldpc_bg_table.h
 *   lib/src/phy/fec/ldpc/ldpc_enc_c.c + ldpc_encoder.c (systematic encoder; restated algebraically
 *                                                 from H, verified against the reference encoder)
 *   lib/src/phy/fec/crc.c:92-140,183-189          (bit-per-byte CRC, zero init, MSB first)
 *
 * The decoder keeps the reference's data model (per-layer check-to-var slab indexed by variable
 * position, v2c computed for the whole high-rate region) so that every intermediate equals the
 * reference's; the HIP kernel uses a compact per-edge layout and is checked against this.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../srslte_amd/csrc/tables/nr_ldpc_bg_table.h"

int orc_ldpc_ls_index(uint16_t ls)
{
  /* base_graph.c:50 LSindex[]: Z = a * 2^j, a in {2,3,5,7,9,11,13,15}, 2 <= Z <= 384 */
  if (ls < 2 || ls > 384) {
    return -1;
  }
  uint16_t odd = ls;
  while ((odd & 1) == 0) {
    odd >>= 1;
  }
  switch (odd) {
    case 1:
      return 0;
    case 3:
      return 1;
    case 5:
      return 2;
    case 7:
      return 3;
    case 9:
      return 4;
    case 11:
      return 5;
    case 13:
      return 6;
    case 15:
      return 7;
    default:
      return -1;
  }
}

int orc_ldpc_graph(orc_ldpc_graph_t* g, int bg, uint16_t ls)
{
  int ils = orc_ldpc_ls_index(ls);
  if (ils < 0 || (bg != 0 && bg != 1)) {
    return -1;
  }
  /* a = 2^j sets exclude Z=1; sets with a=9..15 have max 288,352,208,240; all <= 384 already */
  const nr_ldpc_edge_t* e  = bg == 0 ? nr_ldpc_bg1_edges : nr_ldpc_bg2_edges;
  int                   ne = bg == 0 ? NR_LDPC_BG1_NOF_EDGES : NR_LDPC_BG2_NOF_EDGES;
  memset(g, 0, sizeof(*g));
  g->bg        = bg;
  g->ls        = ls;
  g->bgN       = bg == 0 ? 68 : 52;
  g->bgM       = bg == 0 ? 46 : 42;
  g->bgK       = g->bgN - g->bgM;
  g->nof_edges = ne;
  int row = 0;
  g->row_start[0] = 0;
  for (int i = 0; i < ne; i++) {
    while (row < e[i].row) {
      g->row_start[++row] = (uint16_t)i;
    }
    g->col[i]   = e[i].col;
    g->shift[i] = (uint16_t)(e[i].v[ils] % ls);
  }
  while (row < g->bgM) {
    g->row_start[++row] = (uint16_t)ne;
  }
  return 0;
}

uint32_t orc_crc_bits(uint32_t poly, int order, const uint8_t* bits, int len)
{
  uint64_t mask = ((((uint64_t)1 << (order - 1)) - 1) << 1) | 1;
  uint64_t high = (uint64_t)1 << (order - 1);
  uint64_t crc  = 0;
  for (int i = 0; i < len; i++) {
    uint64_t b   = ((int8_t)bits[i] > 0) ? 1 : 0; /* crc.c:117: signed compare > 0 */
    uint64_t top = (crc & high) ? 1 : 0;
    crc          = (crc << 1) & mask;
    if (top ^ b) {
      crc ^= poly;
    }
  }
  return (uint32_t)(crc & mask);
}

static uint32_t pack_bits(const uint8_t* bits, int n)
{
  uint32_t v = 0;
  for (int i = 0; i < n; i++) {
    v = (v << 1) | (bits[i] & 1); /* bit.c srsran_bit_pack */
  }
  return v;
}

int orc_ldpc_decode_c(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const int8_t* llrs,
                      uint8_t* message, uint32_t cdwd_rm_length, uint32_t crc_poly, int crc_order,
                      int8_t* soft_out)
{
  const int ls = g->ls, bgN = g->bgN, bgM = g->bgM, bgK = g->bgK;
  const int liftN = bgN * ls, liftK = bgK * ls, hrrN = (bgK + 4) * ls;
  if (max_nof_iter == 0) {
    max_nof_iter = 10; /* ldpc_decoder.c:42,579 */
  }
  const int sf = (int)(scaling_fctr * 100); /* ldpc_dec_c.c:150, float * int -> float -> trunc */

  /* ldpc_decoder.c:51-65 */
  if (cdwd_rm_length > (uint32_t)(liftN - 2 * ls)) {
    cdwd_rm_length = liftN - 2 * ls;
  }
  if (cdwd_rm_length < (uint32_t)((bgK + 2) * ls)) {
    cdwd_rm_length = (bgK + 2) * ls;
  }
  if (cdwd_rm_length % ls) {
    cdwd_rm_length = (cdwd_rm_length / ls + 1) * ls;
  }
  const int n_layers = (uint8_t)(cdwd_rm_length / ls - bgK + 2);

  int8_t* soft = malloc(liftN);
  int8_t* c2v  = calloc((size_t)(hrrN + ls) * bgM, 1);
  int8_t* v2c  = calloc(hrrN + ls, 1);
  int8_t (*minv)[2] = malloc(ls * sizeof(int8_t[2]));
  int*    min_idx = calloc(ls, sizeof(int));
  int*    prod    = malloc(ls * sizeof(int));

  /* init_ldpc_dec_c :170-188.  NB: all liftN-2Z llrs are loaded, whatever cdwd_rm_length */
  memset(soft, 0, 2 * ls);
  memcpy(soft + 2 * ls, llrs, liftN - 2 * ls);

  int ret = -2;
  for (int it = 0; it < max_nof_iter && ret == -2; it++) {
    for (int l = 0; l < n_layers; l++) {
      int8_t* this_c2v = c2v + (size_t)l * (hrrN + ls);
      /* update_ldpc_var_to_check_c :190-213 + inner :338-363 */
      for (int i = 0; i < hrrN + (l >= 4 ? ls : 0); i++) {
        int8_t x = (i < hrrN) ? soft[i] : soft[hrrN + (l - 4) * ls + (i - hrrN)];
        if (x >= 127) {
          v2c[i] = 127;
        } else if (x <= -127) {
          v2c[i] = -127;
        } else {
          long t = (long)x - this_c2v[i];
          v2c[i] = (int8_t)(t > 63 ? 63 : (t < -63 ? -63 : t));
        }
      }
      /* update_ldpc_check_to_var_c :215-284 */
      for (int i = 0; i < ls; i++) {
        prod[i]    = 1;
        minv[i][0] = minv[i][1] = INT8_MAX;
      }
      for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {
        int shift = g->shift[e];
        int base  = g->col[e] * ls;
        base      = base <= hrrN ? base : hrrN;
        for (int j = 0; j < ls; j++) {
          int    index = (j + ls - shift) % ls;
          int    iv    = base + j;
          int8_t a     = (int8_t)abs(v2c[iv]);
          int    is_min = a < minv[index][0];
          minv[index][1] = (a >= minv[index][1]) ? minv[index][1] : (is_min ? minv[index][0] : a);
          minv[index][0] = is_min ? a : minv[index][0];
          min_idx[index] = is_min ? iv : min_idx[index];
          prod[index] *= (v2c[iv] >= 0) ? 1 : -1;
        }
      }
      for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {
        int shift = g->shift[e];
        int base  = g->col[e] * ls;
        base      = base <= hrrN ? base : hrrN;
        for (int j = 0; j < ls; j++) {
          int index    = (j + ls - shift) % ls;
          int iv       = base + j;
          this_c2v[iv] = (iv != min_idx[index]) ? minv[index][0] : minv[index][1];
          this_c2v[iv] = (int8_t)(this_c2v[iv] * sf / 100);
          this_c2v[iv] = (int8_t)(this_c2v[iv] * (prod[index] * ((v2c[iv] >= 0) ? 1 : -1)));
        }
      }
      /* update_ldpc_soft_bits_c :286-321 */
      for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {
        int ext = g->col[e] * ls;
        for (int j = 0; j < ls; j++) {
          int  ib  = ext + j;
          int  it2 = (ext <= hrrN) ? ib : hrrN + j;
          long t   = (long)this_c2v[it2] + v2c[it2];
          if (t > 63) {
            t = 127;
          }
          if (t < -63) {
            t = -127;
          }
          soft[ib] = (int8_t)t;
        }
      }
    }
    if (crc_order > 0) {
      for (int i = 0; i < liftK; i++) {
        message[i] = (soft[i] < 0);
      }
      uint32_t c1 = orc_crc_bits(crc_poly, crc_order, message, liftK - crc_order);
      uint32_t c2 = pack_bits(&message[liftK - crc_order], crc_order);
      if (c1 == c2) {
        ret = it + 1;
      }
    }
  }
  if (ret == -2) {
    if (crc_order > 0) {
      ret = 0;
    } else {
      for (int i = 0; i < liftK; i++) {
        message[i] = (soft[i] < 0);
      }
      ret = max_nof_iter;
    }
  }
  if (soft_out) {
    memcpy(soft_out, soft, liftN);
  }
  free(soft);
  free(c2v);
  free(v2c);
  free(minv);
  free(min_idx);
  free(prod);
  return ret;
}

/* ---------------------------------------------------------------- encoder
 * x = [s_0..s_{K-1} | p_0..p_{M-1}] in blocks of Z bits; row m of H:  sum_n P^{shift(m,n)} x_n = 0 with
 * (P^s x)[i] = x[(i+s) mod Z]  (same convention as ldpc_enc_c.c:90-95). */
static void rot_xor(uint8_t* acc, const uint8_t* x, int s, int ls)
{
  for (int i = 0; i < ls; i++) {
    acc[i] ^= x[(i + s) % ls] & 1;
  }
}

int orc_ldpc_encode(const orc_ldpc_graph_t* g, const uint8_t* message, uint8_t* codeword)
{
  const int ls = g->ls, bgN = g->bgN, bgM = g->bgM, bgK = g->bgK;
  uint8_t*  x     = calloc((size_t)bgN * ls, 1);
  uint8_t*  lam   = calloc((size_t)4 * ls, 1);
  int       known[4] = {0, 0, 0, 0};
  for (int i = 0; i < bgK * ls; i++) {
    x[i] = message[i] & 1; /* filler flag (254) & 1 = 0, ldpc_enc_c.c:93 */
  }
  /* lambda_m = systematic part of core rows 0..3 */
  for (int m = 0; m < 4; m++) {
    for (int e = g->row_start[m]; e < g->row_start[m + 1]; e++) {
      if (g->col[e] < bgK) {
        rot_xor(&lam[m * ls], &x[g->col[e] * ls], g->shift[e], ls);
      }
    }
  }
  /* p0: adding the four core rows cancels the dual-diagonal columns and two of the three entries of
   * column bgK (equal shifts); one rotation of p0 remains */
  {
    int cnt[384 + 1];
    int shifts[4], ns = 0;
    memset(cnt, 0, sizeof(cnt));
    for (int m = 0; m < 4; m++) {
      for (int e = g->row_start[m]; e < g->row_start[m + 1]; e++) {
        if (g->col[e] == bgK) {
          shifts[ns++] = g->shift[e];
        }
      }
    }
    int a = -1;
    for (int i = 0; i < ns; i++) {
      int c = 0;
      for (int j = 0; j < ns; j++) {
        c += shifts[j] == shifts[i];
      }
      if (c % 2 == 1) {
        a = shifts[i];
      }
    }
    if (a < 0) {
      free(x);
      free(lam);
      return -1;
    }
    uint8_t* p0 = &x[bgK * ls];
    for (int k = 0; k < ls; k++) {
      uint8_t s = lam[k] ^ lam[ls + k] ^ lam[2 * ls + k] ^ lam[3 * ls + k];
      p0[(k + a) % ls] = s;
    }
    known[0] = 1;
  }
  /* remaining core parity blocks: repeatedly pick a core row with exactly one unknown parity block */
  for (int pass = 0; pass < 4; pass++) {
    for (int m = 0; m < 4; m++) {
      int unk = -1, nunk = 0, ushift = 0;
      for (int e = g->row_start[m]; e < g->row_start[m + 1]; e++) {
        int c = g->col[e];
        if (c >= bgK && c < bgK + 4 && !known[c - bgK]) {
          unk = c;
          ushift = g->shift[e];
          nunk++;
        }
      }
      if (nunk != 1) {
        continue;
      }
      uint8_t* rhs = calloc(ls, 1);
      memcpy(rhs, &lam[m * ls], ls);
      for (int e = g->row_start[m]; e < g->row_start[m + 1]; e++) {
        int c = g->col[e];
        if (c >= bgK && c < bgK + 4 && c != unk) {
          rot_xor(rhs, &x[c * ls], g->shift[e], ls);
        }
      }
      for (int k = 0; k < ls; k++) {
        x[unk * ls + (k + ushift) % ls] = rhs[k];
      }
      free(rhs);
      known[unk - bgK] = 1;
    }
  }
  if (!(known[1] && known[2] && known[3])) {
    free(x);
    free(lam);
    return -1;
  }
  /* extension rows: p_m = sum of everything else in the row (its own column has shift 0) */
  for (int m = 4; m < bgM; m++) {
    uint8_t* p = &x[(bgK + m) * ls];
    for (int e = g->row_start[m]; e < g->row_start[m + 1]; e++) {
      if (g->col[e] < bgK + 4) {
        rot_xor(p, &x[g->col[e] * ls], g->shift[e], ls);
      }
    }
  }
  memcpy(codeword, &x[2 * ls], (size_t)(bgN - 2) * ls);
  free(x);
  free(lam);
  return 0;
}

/* ---------------------------------------------------------------- encoder with the reference's output conventions
 * ldpc_encoder.c:55-95 (encode_c): the systematic part is the raw input (filler flags kept), only the parity blocks of the
 * first n_layers check rows are written, n_layers follows from cdwd_rm_length after its clamping / rounding; everything
 * beyond is left untouched. */
int orc_ldpc_encode_rm(const orc_ldpc_graph_t* g, const uint8_t* input, uint8_t* output, uint32_t cdwd_rm_length)
{
  const uint32_t ls = g->ls, full = (uint32_t)(g->bgN - 2) * ls;
  if (cdwd_rm_length > full) {
    cdwd_rm_length = full;
  }
  if (cdwd_rm_length < (uint32_t)(g->bgK + 2) * ls) {
    cdwd_rm_length = (uint32_t)(g->bgK + 2) * ls;
  }
  if (cdwd_rm_length % ls) {
    cdwd_rm_length = (cdwd_rm_length / ls + 1) * ls;
  }
  uint8_t* cw = malloc(full);
  if (!cw || orc_ldpc_encode(g, input, cw) != 0) {
    free(cw);
    return -1;
  }
  memcpy(output, input + 2 * ls, (size_t)(g->bgK - 2) * ls);
  memcpy(output + (size_t)(g->bgK - 2) * ls, cw + (size_t)(g->bgK - 2) * ls, cdwd_rm_length - (size_t)(g->bgK - 2) * ls);
  free(cw);
  return 0;
}

/* ---------------------------------------------------------------- rate matching, ldpc_rm.c
 * init_rm (ldpc_rm.c:113-167): N = 66 Z / 50 Z (the codeword without the two punctured blocks), starting positions
 * k0 = Z * BASEK0[rv] (scaled when the circular buffer is limited to Nref), filler positions [K - 2Z - F, K - 2Z). */
static int rm_params(int bg, uint32_t ls, uint32_t rv, uint32_t Nref, uint32_t* N, uint32_t* K, uint32_t* Ncb, uint32_t* k0)
{
  static const uint32_t basek0[4][2] = {{0, 0}, {17, 13}, {33, 25}, {56, 43}}; /* TS 38.212 table 5.4.2.1-2 */
  if (bg < 0 || bg > 1 || rv > 3) {
    return -1;
  }
  *N = ls * (bg == 0 ? 66 : 50);
  *K = ls * (bg == 0 ? 22 : 10);
  if (*N <= Nref) {
    *Ncb = *N;
    *k0  = ls * basek0[rv][bg];
  } else {
    *Ncb = Nref;
    *k0  = ls * ((basek0[rv][bg] * Nref) / *N);
  }
  return 0;
}

/* srsran_ldpc_rm_tx (ldpc_rm.c:173-193,348-362,582-610): select E non-filler bits from k0 on, then interleave */
int orc_ldpc_rm_tx(const uint8_t* input, uint8_t* output, uint32_t E, int bg, uint32_t ls, uint32_t rv, uint32_t Qm, uint32_t Nref)
{
  uint32_t N, K, Ncb, k0;
  if (rm_params(bg, ls, rv, Nref, &N, &K, &Ncb, &k0) || Qm == 0 || E % Qm) {
    return -1;
  }
  uint8_t* tmp = malloc(E ? E : 1);
  for (uint32_t k = 0, j = 0; k < E; j++) {
    uint32_t i = (k0 + j) % Ncb;
    if (input[i] != 254) {
      tmp[k++] = input[i];
    }
  }
  const uint32_t cols = E / Qm;
  for (uint32_t j = 0; j < cols; j++) {
    for (uint32_t i = 0; i < Qm; i++) {
      output[i + j * Qm] = tmp[i * cols + j];
    }
  }
  free(tmp);
  return 0;
}

/* srsran_ldpc_rm_rx_{c,s,f} (ldpc_rm.c:203-346,365-411,612-706): de-interleave, mark fillers as "infinity", accumulate
 * the E soft bits in transmission order with saturation at +-63 / +-16383 (none for float).  type: 0 int8, 1 int16,
 * 2 float.  Returns min(k0 + E, Ncb) like the int8 function (the other two return 0 in the reference). */
int orc_ldpc_rm_rx(int type, const void* input, void* output, uint32_t E, uint32_t F, int bg, uint32_t ls, uint32_t rv, uint32_t Qm,
                   uint32_t Nref)
{
  uint32_t N, K, Ncb, k0;
  if (rm_params(bg, ls, rv, Nref, &N, &K, &Ncb, &k0) || Qm == 0 || E % Qm) {
    return -1;
  }
  const uint32_t end_ex = K - 2 * ls, ini_ex = end_ex - F, cols = E / Qm;
  for (uint32_t i = ini_ex; i < end_ex; i++) {
    if (type == 0) {
      ((int8_t*)output)[i] = 127;
    } else if (type == 1) {
      ((int16_t*)output)[i] = 32767;
    } else {
      ((float*)output)[i] = INFINITY;
    }
  }
  for (uint32_t k = 0, j = 0; k < E; j++) {
    uint32_t idx = (k0 + j) % Ncb;
    if (idx >= ini_ex && idx < end_ex) {
      continue;
    }
    const uint32_t src = Qm == 1 ? k : (k % cols) * Qm + k / cols; /* tmp[i*cols + j] = in[j*Qm + i] */
    if (type == 0) {
      long t = (long)((int8_t*)output)[idx] + ((const int8_t*)input)[src];
      t      = t > 63 ? 63 : (t < -63 ? -63 : t);
      ((int8_t*)output)[idx] = (int8_t)t;
    } else if (type == 1) {
      long t = (long)((int16_t*)output)[idx] + ((const int16_t*)input)[src];
      t      = t > 16383 ? 16383 : (t < -16383 ? -16383 : t);
      ((int16_t*)output)[idx] = (int16_t)t;
    } else {
      ((float*)output)[idx] = ((float*)output)[idx] + ((const float*)input)[src];
    }
    k++;
  }
  return (int)(k0 + E < Ncb ? k0 + E : Ncb);
}

/* ---------------------------------------------------------------- flooded schedule, int8
 * ldpc_decoder.c:105-160 (LDPC_DECODER_TEMPLATE_FLOOD) with ldpc_dec_c_flood.c: per iteration all variable-to-check
 * messages (from the soft bits of the previous iteration), then all check-to-variable messages, then the soft bits are
 * rebuilt from the channel LLRs by adding the messages of ALL bgM rows in row order with the +-63 -> +-127 saturation
 * after every addition (rows beyond n_layers add zero, the saturation still applies).  2 * max_nof_iter iterations. */
int orc_ldpc_decode_c_flood(const orc_ldpc_graph_t* g, float scaling_fctr, int max_nof_iter, const int8_t* llrs,
                            uint8_t* message, uint32_t cdwd_rm_length, uint32_t crc_poly, int crc_order, int8_t* soft_out)
{
  const int ls = g->ls, bgN = g->bgN, bgM = g->bgM, bgK = g->bgK;
  const int liftN = bgN * ls, liftK = bgK * ls, hrrN = (bgK + 4) * ls, W = hrrN + ls;
  if (max_nof_iter == 0) {
    max_nof_iter = 10;
  }
  const int sf = (int)(scaling_fctr * 100); /* ldpc_dec_c_flood.c: scaling_fctr * F2I */
  if (cdwd_rm_length > (uint32_t)(liftN - 2 * ls)) {
    cdwd_rm_length = liftN - 2 * ls;
  }
  if (cdwd_rm_length < (uint32_t)((bgK + 2) * ls)) {
    cdwd_rm_length = (bgK + 2) * ls;
  }
  if (cdwd_rm_length % ls) {
    cdwd_rm_length = (cdwd_rm_length / ls + 1) * ls;
  }
  const int n_layers = (uint8_t)(cdwd_rm_length / ls - bgK + 2);

  int8_t* llr  = malloc(liftN);
  int8_t* soft = malloc(liftN);
  int8_t* c2v  = calloc((size_t)W * bgM, 1);
  int8_t* v2c  = calloc((size_t)W * bgM, 1);
  int8_t (*minv)[2] = malloc(ls * sizeof(int8_t[2]));
  int*    min_idx = calloc(ls, sizeof(int)); /* uninitialised in the reference; only read when it was just written */
  int*    prod    = malloc(ls * sizeof(int));
  memset(llr, 0, 2 * ls);
  memcpy(llr + 2 * ls, llrs, liftN - 2 * ls);
  memcpy(soft, llr, liftN);

  int ret = -2;
  for (int it = 0; it < 2 * max_nof_iter && ret == -2; it++) {
    for (int l = 0; l < n_layers; l++) { /* ldpc_dec_c_flood.c:205-229 + :366-391 */
      int8_t* tc = c2v + (size_t)l * W;
      int8_t* tv = v2c + (size_t)l * W;
      for (int i = 0; i < hrrN + (l >= 4 ? ls : 0); i++) {
        int8_t x = (i < hrrN) ? soft[i] : soft[hrrN + (l - 4) * ls + (i - hrrN)];
        if (x >= 127) {
          tv[i] = 127;
        } else if (x <= -127) {
          tv[i] = -127;
        } else {
          long t = (long)x - tc[i];
          tv[i]  = (int8_t)(t > 63 ? 63 : (t < -63 ? -63 : t));
        }
      }
    }
    for (int l = 0; l < n_layers; l++) { /* :231-302 */
      int8_t* tc = c2v + (size_t)l * W;
      int8_t* tv = v2c + (size_t)l * W;
      for (int i = 0; i < ls; i++) {
        prod[i]    = 1;
        minv[i][0] = minv[i][1] = INT8_MAX;
      }
      for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {
        int base = g->col[e] * ls;
        base     = base <= hrrN ? base : hrrN;
        for (int j = 0; j < ls; j++) {
          int    index  = (j + ls - g->shift[e]) % ls;
          int    iv     = base + j;
          int8_t a      = (int8_t)abs(tv[iv]);
          int    is_min = a < minv[index][0];
          minv[index][1] = (a >= minv[index][1]) ? minv[index][1] : (is_min ? minv[index][0] : a);
          minv[index][0] = is_min ? a : minv[index][0];
          min_idx[index] = is_min ? iv : min_idx[index];
          prod[index] *= (tv[iv] >= 0) ? 1 : -1;
        }
      }
      for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {
        int base = g->col[e] * ls;
        base     = base <= hrrN ? base : hrrN;
        for (int j = 0; j < ls; j++) {
          int index = (j + ls - g->shift[e]) % ls;
          int iv    = base + j;
          tc[iv]    = (iv != min_idx[index]) ? minv[index][0] : minv[index][1];
          tc[iv]    = (int8_t)(tc[iv] * sf / 100);
          tc[iv]    = (int8_t)(tc[iv] * (prod[index] * ((tv[iv] >= 0) ? 1 : -1)));
        }
      }
    }
    memcpy(soft, llr, liftN); /* :304-349 */
    for (int l = 0; l < bgM; l++) {
      const int8_t* tc = c2v + (size_t)l * W;
      for (int e = g->row_start[l]; e < g->row_start[l + 1]; e++) {
        int ext = g->col[e] * ls;
        for (int j = 0; j < ls; j++) {
          int  ib = ext + j;
          int  ic = (ext <= hrrN) ? ib : hrrN + j;
          long t  = (long)tc[ic] + soft[ib];
          if (t > 63) {
            t = INT8_MAX;
          }
          if (t < -63) {
            t = -INT8_MAX;
          }
          soft[ib] = (int8_t)t;
        }
      }
    }
    if (crc_order > 0) {
      for (int i = 0; i < liftK; i++) {
        message[i] = (soft[i] < 0);
      }
      uint32_t c1 = orc_crc_bits(crc_poly, crc_order, message, liftK - crc_order);
      uint32_t c2 = pack_bits(&message[liftK - crc_order], crc_order);
      if (c1 == c2) {
        ret = it + 1;
      }
    }
  }
  if (ret == -2) {
    if (crc_order > 0) {
      ret = 0;
    } else {
      for (int i = 0; i < liftK; i++) {
        message[i] = (soft[i] < 0);
      }
      ret = max_nof_iter;
    }
  }
  if (soft_out) {
    memcpy(soft_out, soft, liftN);
  }
  free(llr);
  free(soft);
  free(c2v);
  free(v2c);
  free(minv);
  free(min_idx);
  free(prod);
  return ret;
}
