/* orc_sch_nr.c -- TEST INFRASTRUCTURE (see oracle/README.md): CPU restatement of the NR shared-channel transport-block
 * loop of the reference, lib/src/phy/phch/sch_nr.c (segmentation: lib/src/phy/fec/cbsegm.c:159-285), built on the
 * oracle's LDPC encoder / decoder / rate matcher.  Not part of the product.
 *
 * Pinning: sch_nr.c itself cannot be linked here (its srsran_ra_nr_tbs pulls the reference's whole NR control plane), so
 * the fixtures for this file are made by tools/gen_golden.py from the reference's own building blocks in oracle/_ref
 * (srsran_cbsegm_ldpc_bg1/2, srsran_crc_*, srsran_ldpc_encoder_encode, srsran_ldpc_rm_tx / _rx_c,
 * srsran_ldpc_decoder_decode_crc_c, srsran_bit_*) driven in the order sch_nr.c calls them.  The one input taken from the
 * caller instead of being derived is Nref (sch_nr.c:119-126 computes it from the carrier through srsran_ra_nr_tbs). */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CRC24A 0x1864CFBu
#define CRC24B 0x1800063u
#define CRC16 0x11021u
#define FILLER 254

/* srsran_sch_nr_select_basegraph, sch_nr.c:35-45 */
int orc_sch_nr_select_basegraph(uint32_t tbs, double R)
{
  return ((tbs <= 292) || (tbs <= 3824 && R <= 0.67) || (R <= 0.25)) ? 1 : 0;
}

/* srsran_cbsegm_ldpc, cbsegm.c:207-275 (with cbsegm_cb_size :51-60 and cbsegm_ldpc_select_ls :159-191) */
static int cbsegm_ldpc(int bg, uint32_t tbs, uint32_t* C, uint32_t* K, uint32_t* Z, uint32_t* L_tb, uint32_t* L_cb)
{
  const uint32_t L    = tbs <= 3824 ? 16 : 24;
  const uint32_t K_cb = bg == 0 ? 8448 : 3840;
  const uint32_t B    = tbs + L;
  uint32_t       c, Bp;
  if (B <= K_cb) {
    c  = 1;
    Bp = B;
  } else {
    c  = (B + (K_cb - 24) - 1) / (K_cb - 24);
    Bp = B + 24 * c;
  }
  const uint32_t Kp  = Bp / c;
  uint32_t       K_b = 22;
  if (bg == 1) {
    K_b = B > 640 ? 10 : (B > 560 ? 9 : (B > 192 ? 8 : 6));
  }
  uint32_t z = (Kp + K_b - 1) / K_b;
  for (; z <= 384; z++) {
    if (orc_ldpc_ls_index((uint16_t)z) >= 0) {
      break;
    }
  }
  if (z > 384) {
    return -1;
  }
  *C    = c;
  *Z    = z;
  *K    = z * (bg == 0 ? 22u : 10u);
  *L_tb = L;
  *L_cb = c > 1 ? 24 : 0;
  return 0;
}

/* srsran_sch_nr_fill_tb_info, sch_nr.c:77-144, Nref handed in */
int orc_sch_nr_tb_info(uint32_t tbs, double R, uint32_t Qm, uint32_t nof_bits, uint32_t N_L, uint32_t Nref, orc_nr_tb_info_t* cfg)
{
  uint32_t C, K, Z, L_tb, L_cb;
  const int bg = orc_sch_nr_select_basegraph(tbs, R);
  if (tbs == 0 || cbsegm_ldpc(bg, tbs, &C, &K, &Z, &L_tb, &L_cb)) {
    return -1;
  }
  cfg->bg   = (uint32_t)bg;
  cfg->Qm   = Qm;
  cfg->A    = tbs;
  cfg->L_tb = L_tb;
  cfg->L_cb = L_cb;
  cfg->B    = tbs + L_tb;
  cfg->Bp   = cfg->B + L_cb * C;
  cfg->Kp   = cfg->Bp / C;
  cfg->Kr   = K;
  cfg->F    = cfg->Kr - cfg->Kp;
  cfg->Z    = Z;
  cfg->G    = nof_bits;
  cfg->Nl   = N_L;
  cfg->Nref = Nref;
  cfg->C    = C;
  return 0;
}

/* sch_nr_get_E, sch_nr.c:146-157 (all code blocks transmitted: Cp = C) */
uint32_t orc_sch_nr_get_E(const orc_nr_tb_info_t* cfg, uint32_t j)
{
  if (cfg->Nl == 0 || cfg->Qm == 0 || cfg->C == 0) {
    return 0;
  }
  const uint32_t q = cfg->Nl * cfg->Qm;
  if (j <= (cfg->C - (cfg->G / q) % cfg->C - 1)) {
    return q * (cfg->G / (q * cfg->C));
  }
  return q * ((cfg->G + q * cfg->C - 1) / (q * cfg->C));
}

static uint32_t crc_of(uint32_t L)
{
  return L == 24 ? CRC24A : CRC16;
}

/* sch_nr_encode, sch_nr.c:375-520: data = tbs / 8 bytes, e_bits = G bytes (one bit each) */
int orc_sch_nr_encode_tb(const orc_nr_tb_info_t* cfg, uint32_t rv, const uint8_t* data, uint8_t* e_bits)
{
  orc_ldpc_graph_t g;
  if (orc_ldpc_graph(&g, (int)cfg->bg, (uint16_t)cfg->Z)) {
    return -1;
  }
  const uint32_t N = (uint32_t)(g.bgN - 2) * cfg->Z;
  uint8_t*       tb_bits = malloc(cfg->A + 24), *cb = malloc(cfg->Kr), *cw = malloc(N);
  for (uint32_t i = 0; i < cfg->A; i++) {
    tb_bits[i] = (data[i / 8] >> (7 - i % 8)) & 1;
  }
  const uint32_t checksum_tb = orc_crc_bits(crc_of(cfg->L_tb), (int)cfg->L_tb, tb_bits, (int)cfg->A);
  uint32_t       in = 0, out = 0;
  for (uint32_t r = 0; r < cfg->C; r++) {
    uint32_t cb_len = cfg->Kp - cfg->L_cb;
    if (r == cfg->C - 1) {
      cb_len -= cfg->L_tb;
      memcpy(cb, tb_bits + in, cb_len);
      for (uint32_t i = 0; i < cfg->L_tb; i++) {
        cb[cb_len + i] = (checksum_tb >> (cfg->L_tb - 1 - i)) & 1;
      }
    } else {
      memcpy(cb, tb_bits + in, cb_len);
    }
    in += (cb_len / 8) * 8; /* input_ptr += cb_len / 8 (bytes) */
    if (cfg->L_cb) {
      const uint32_t n = cfg->Kp - cfg->L_cb, c = orc_crc_bits(CRC24B, 24, cb, (int)n);
      for (uint32_t i = 0; i < 24; i++) {
        cb[n + i] = (c >> (23 - i)) & 1;
      }
    }
    for (uint32_t i = cfg->Kp; i < cfg->Kr; i++) {
      cb[i] = FILLER;
    }
    orc_ldpc_encode_rm(&g, cb, cw, N); /* keeps the filler marks in the systematic part: the rate matcher skips them by value */
    const uint32_t E = orc_sch_nr_get_E(cfg, r);
    orc_ldpc_rm_tx(cw, e_bits + out, E, (int)cfg->bg, cfg->Z, rv, cfg->Qm, cfg->Nref);
    out += E;
  }
  free(tb_bits);
  free(cb);
  free(cw);
  return 0;
}

/* sch_nr_decode, sch_nr.c:522-713.
 *   e_bits   : rate-matched LLRs of the code blocks that are still undecoded, back to back (:665 advances only for those)
 *   softbuf  : C x sb_stride int8, accumulated across transmissions;  cb_crc: C flags, in / out
 *   cb_data  : C x data_stride bytes, the packed bits of decoded code blocks (softbuffer.rx->data)
 *   payload  : tbs / 8 bytes, written when every code block is decoded;  crc_ok: 1 / 0 (left 0 when not all decoded) */
int orc_sch_nr_decode_tb(const orc_nr_tb_info_t* cfg, uint32_t rv, float scaling_fctr, int max_nof_iter, const int8_t* e_bits,
                         int8_t* softbuf, uint32_t sb_stride, uint8_t* cb_crc, uint8_t* cb_data, uint32_t data_stride, uint8_t* payload,
                         int* crc_ok, float* avg_iter)
{
  orc_ldpc_graph_t g;
  if (orc_ldpc_graph(&g, (int)cfg->bg, (uint16_t)cfg->Z)) {
    return -1;
  }
  const uint32_t liftK = (uint32_t)g.bgK * cfg->Z;
  uint8_t*       temp  = malloc(liftK);
  uint32_t       cb_ok = 0, nof_iter_sum = 0, in = 0;
  *crc_ok              = 0;
  for (uint32_t r = 0; r < cfg->C; r++) {
    const uint32_t E = orc_sch_nr_get_E(cfg, r);
    if (cb_crc[r]) {
      cb_ok++;
      continue;
    }
    int8_t*   rm    = softbuf + (size_t)r * sb_stride;
    const int n_llr = orc_ldpc_rm_rx(0, e_bits + in, rm, E, cfg->F, (int)cfg->bg, cfg->Z, rv, cfg->Qm, cfg->Nref);
    if (n_llr < 0) {
      free(temp);
      return -1;
    }
    uint32_t poly = crc_of(cfg->L_tb), order = cfg->L_tb;
    if (cfg->L_cb) {
      poly  = CRC24B;
      order = 24;
    }
    const int ret = orc_ldpc_decode_c(&g, scaling_fctr, max_nof_iter, rm, temp, (uint32_t)n_llr, poly, (int)order, NULL);
    if (ret < 0) {
      free(temp);
      return -1;
    }
    nof_iter_sum += ret == 0 ? (uint32_t)max_nof_iter : (uint32_t)ret;
    const uint32_t cb_len = cfg->Kp - cfg->L_cb;
    int            all_zeros = 1;
    for (uint32_t i = 0; i < cb_len && all_zeros; i++) {
      all_zeros = temp[i] == 0;
    }
    cb_crc[r] = (ret != 0) && !all_zeros;
    if (cb_crc[r]) {
      uint8_t* d = cb_data + (size_t)r * data_stride;
      memset(d, 0, (cb_len + 7) / 8);
      for (uint32_t i = 0; i < cb_len; i++) {
        d[i / 8] |= (uint8_t)(temp[i] << (7 - i % 8));
      }
      cb_ok++;
    }
    in += E;
  }
  free(temp);
  *avg_iter = cfg->C ? (float)nof_iter_sum / (float)cfg->C : NAN;
  if (cb_ok != cfg->C) {
    return 0;
  }
  uint32_t checksum2 = 0, out = 0;
  for (uint32_t r = 0; r < cfg->C; r++) {
    uint32_t cb_len = cfg->Kp - cfg->L_cb;
    if (r == cfg->C - 1) {
      cb_len -= cfg->L_tb;
    }
    const uint8_t* d = cb_data + (size_t)r * data_stride;
    memcpy(payload + out, d, cb_len / 8);
    out += cb_len / 8;
    if (cfg->C > 1 && r == cfg->C - 1) {
      for (uint32_t i = 0; i < cfg->L_tb; i++) {
        const uint32_t b = cb_len + i; /* srsran_bit_unpack_vector from byte cb_len / 8 on */
        checksum2        = (checksum2 << 1) | ((d[b / 8] >> (7 - b % 8)) & 1u);
      }
    }
  }
  if (cfg->C == 1) {
    *crc_ok = 1;
  } else {
    uint8_t* bits = malloc(cfg->A);
    for (uint32_t i = 0; i < cfg->A; i++) {
      bits[i] = (payload[i / 8] >> (7 - i % 8)) & 1;
    }
    *crc_ok = orc_crc_bits(crc_of(cfg->L_tb), (int)cfg->L_tb, bits, (int)cfg->A) == checksum2;
    free(bits);
  }
  return 0;
}
