/* tests/ref_link/chan_bind.c -- the reference-side binding of the grant-level entry points.
 *
 * Compiled against the REFERENCE's headers, like a file a maintainer adds to lib/src/phy/phch.  pusch.c, pdsch.c and sch.c are compiled
 * unmodified; in their OBJECT files the definitions of srsran_pusch_decode, srsran_pdsch_decode, srsran_pdsch_encode and srsran_ulsch_encode are
 * renamed to <name>_ref (objcopy --redefine-sym, tests/ref_link/Makefile) and the functions below take the original names: every caller in
 * the reference (srsran_enb_ul_get_pusch, srsran_ue_dl_decode_pdsch, srsran_enb_dl_put_pdsch, srsran_pusch_encode, the test programs) reaches
 * them.  A grant the device path takes costs ONE host <-> device round trip (include/srsran_amd/phy_chan_abi.h); what it does not take -- UCI
 * on PUSCH, more than one port / antenna / codeword or CSI weighting on PDSCH receive, EVM measurement -- goes to the renamed original.
 * In a source integration these are the bodies of the four functions behind an `if (device_takes(cfg))`. */
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "srsran/phy/ch_estimation/chest_ul.h"
#include "srsran/phy/mimo/layermap.h"
#include "srsran/phy/mimo/precoding.h"
#include "srsran/phy/phch/pdsch.h"
#include "srsran/phy/phch/pusch.h"
#include "srsran/phy/phch/sch.h"
#include "srsran/phy/utils/debug.h"
#include "srsran/phy/utils/vector.h"

/* include/srsran_amd/phy_chan_abi.h (declared by hand: its struct names for the soft buffers are the reference's own) */
typedef struct {
  uint32_t mod, tbs, rv, nof_re, seed, max_nof_iterations, llr_is_8bit, nl;
} srsran_hip_grant_tb_t;
typedef struct {
  int32_t crc_ok;
  float   avg_iterations_block, epre;
} srsran_hip_grant_res_t;
typedef struct {
  srsran_hip_grant_tb_t tb;
  uint32_t              cell_nof_prb, cp_nsymb, n_prb_tilde[2], L_prb, shortened;
  float                 noise_estimate;
  uint32_t              meas_epre;
} srsran_hip_pusch_rx_t;
typedef struct {
  srsran_hip_grant_tb_t tb;
  float                 scaling, noise_estimate;
} srsran_hip_pdsch_rx_t;
typedef struct {
  srsran_hip_grant_tb_t tb;
  float                 scaling;
} srsran_hip_pdsch_tx_t;
extern int      srsran_hip_pusch_decode(const srsran_hip_pusch_rx_t*, const cf_t*, const cf_t*, srsran_softbuffer_rx_t*, uint8_t*, srsran_hip_grant_res_t*);
extern int      srsran_hip_pdsch_decode_dbg(const srsran_hip_pdsch_rx_t*, const cf_t*, const cf_t*, srsran_softbuffer_rx_t*, uint8_t*, srsran_hip_grant_res_t*, cf_t*, void*);
extern int      srsran_hip_pdsch_encode_dbg(const srsran_hip_pdsch_tx_t*, srsran_softbuffer_tx_t*, uint8_t*, cf_t*, uint8_t*);
extern int      srsran_hip_ulsch_encode(const srsran_hip_grant_tb_t*, uint32_t, srsran_softbuffer_tx_t*, uint8_t*, uint8_t*);
extern uint32_t srsran_hip_sequence_pusch_seed(uint16_t rnti, uint32_t nslot, uint32_t cell_id);
extern uint32_t srsran_hip_sequence_pdsch_seed(uint16_t rnti, int q, uint32_t nslot, uint32_t cell_id);

/* the originals, renamed in the object files */
extern int srsran_pusch_decode_ref(srsran_pusch_t*, srsran_ul_sf_cfg_t*, srsran_pusch_cfg_t*, srsran_chest_ul_res_t*, cf_t*, srsran_pusch_res_t*);
extern int srsran_pdsch_decode_ref(srsran_pdsch_t*, srsran_dl_sf_cfg_t*, srsran_pdsch_cfg_t*, srsran_chest_dl_res_t*, cf_t* [SRSRAN_MAX_PORTS], srsran_pdsch_res_t [SRSRAN_MAX_CODEWORDS]);
extern int srsran_pdsch_encode_ref(srsran_pdsch_t*, srsran_dl_sf_cfg_t*, srsran_pdsch_cfg_t*, uint8_t* [SRSRAN_MAX_CODEWORDS], cf_t* [SRSRAN_MAX_PORTS]);
extern int srsran_ulsch_encode_ref(srsran_sch_t*, srsran_pusch_cfg_t*, uint8_t*, srsran_uci_value_t*, uint8_t*, uint8_t*);

/* how many grants went which way (read by the tests through the programs' exit: CHAN_BIND_REPORT=1 prints them at exit) */
static unsigned n_dev[4], n_ref[4];
static void     report(void)
{
  fprintf(stderr,
          "[chan_bind] pusch_decode dev %u ref %u | pdsch_decode dev %u ref %u | pdsch_encode dev %u ref %u | ulsch_encode dev %u ref %u\n",
          n_dev[0], n_ref[0], n_dev[1], n_ref[1], n_dev[2], n_ref[2], n_dev[3], n_ref[3]);
}
static void count(int which, bool dev)
{
  static bool armed;
  if (!armed) {
    armed = true;
    if (getenv("CHAN_BIND_REPORT")) {
      atexit(report);
    }
  }
  (dev ? n_dev : n_ref)[which]++;
}

static bool has_uci(const srsran_pusch_cfg_t* cfg)
{
  return srsran_uci_cfg_total_ack(&cfg->uci_cfg) > 0 || cfg->uci_cfg.cqi.ri_len > 0 || cfg->uci_cfg.cqi.data_enable;
}

static uint32_t elapsed_us(const struct timeval* a)
{
  struct timeval b;
  gettimeofday(&b, NULL);
  return (uint32_t)((b.tv_sec - a->tv_sec) * 1000000L + (b.tv_usec - a->tv_usec));
}

/* ---- pusch.c:358 */
int srsran_pusch_decode(srsran_pusch_t* q, srsran_ul_sf_cfg_t* sf, srsran_pusch_cfg_t* cfg, srsran_chest_ul_res_t* channel, cf_t* sf_symbols, srsran_pusch_res_t* out)
{
  if (!q || !sf_symbols || !out || !cfg || !sf || !channel) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  const bool evm = cfg->meas_evm_en && q->evm_buffer;
  if (has_uci(cfg) || cfg->grant.tb.tbs <= 0 || evm || !cfg->softbuffers.rx || !out->data) {
    count(0, false);
    return srsran_pusch_decode_ref(q, sf, cfg, channel, sf_symbols, out);
  }
  count(0, true);
  struct timeval t0;
  if (cfg->meas_time_en) {
    gettimeofday(&t0, NULL);
  }
  if (!cfg->enable_64qam && cfg->grant.tb.mod >= SRSRAN_MOD_64QAM) { /* pusch.c:374-380 */
    cfg->grant.tb.mod      = SRSRAN_MOD_16QAM;
    cfg->grant.tb.nof_bits = cfg->grant.nof_re * srsran_mod_bits_x_symbol(SRSRAN_MOD_16QAM);
  }
  const uint32_t        nslot = 2 * (sf->tti % SRSRAN_NOF_SF_X_FRAME);
  srsran_hip_pusch_rx_t g     = {.tb             = {.mod                = (uint32_t)cfg->grant.tb.mod,
                                                    .tbs                = (uint32_t)cfg->grant.tb.tbs,
                                                    .rv                 = (uint32_t)cfg->grant.tb.rv,
                                                    .nof_re             = cfg->grant.nof_re,
                                                    .seed               = srsran_hip_sequence_pusch_seed(cfg->rnti, nslot, q->cell.id),
                                                    .max_nof_iterations = 0, /* below */
                                                    .llr_is_8bit        = q->llr_is_8bit,
                                                    .nl                 = 1},
                                 .cell_nof_prb   = q->cell.nof_prb,
                                 .cp_nsymb       = SRSRAN_CP_NSYMB(q->cell.cp),
                                 .n_prb_tilde    = {cfg->grant.n_prb_tilde[0], cfg->grant.n_prb_tilde[1]},
                                 .L_prb          = cfg->grant.L_prb,
                                 .shortened      = sf->shortened,
                                 .noise_estimate = channel->noise_estimate,
                                 .meas_epre      = cfg->meas_epre_en};
  srsran_hip_grant_res_t r;
  srsran_sch_set_max_noi(&q->ul_sch, cfg->max_nof_iterations); /* :452 (0 = the reference's default) */
  g.tb.max_nof_iterations = q->ul_sch.max_iterations;
  srsran_cbsegm_t seg;
  if (srsran_cbsegm(&seg, g.tb.tbs) == SRSRAN_SUCCESS) {
    cfg->K_segm = seg.C1 * seg.K1 + seg.C2 * seg.K2; /* sch.c:1141 */
  }
  if (srsran_hip_pusch_decode(&g, sf_symbols, channel->ce, cfg->softbuffers.rx, out->data, &r) != SRSRAN_SUCCESS) {
    return SRSRAN_ERROR;
  }
  out->crc                  = r.crc_ok != 0;
  out->avg_iterations_block = r.avg_iterations_block;
  q->ul_sch.avg_iterations  = r.avg_iterations_block;
  out->epre_dbfs            = cfg->meas_epre_en ? srsran_convert_power_to_dB(r.epre) : NAN;
  out->evm                  = NAN;
  cfg->last_O_cqi           = srsran_cqi_size(&cfg->uci_cfg.cqi); /* :462 */
  if (cfg->meas_time_en) {
    cfg->meas_time_value = elapsed_us(&t0);
  }
  return SRSRAN_SUCCESS;
}

/* ---- pdsch.c:788.  Device path: one port, one receive antenna, one codeword on one layer, no CSI weighting, no EVM, and a power allocation
 * whose rho_b leaves the reference symbols alone (pdsch.c:486-519: p_b = 0 on one port); the resource extraction stays the reference's
 * srsran_pdsch_get (cell-specific reference signals, PSS / SSS / PBCH holes). */
int srsran_pdsch_decode(srsran_pdsch_t* q, srsran_dl_sf_cfg_t* sf, srsran_pdsch_cfg_t* cfg, srsran_chest_dl_res_t* channel, cf_t* sf_symbols[SRSRAN_MAX_PORTS],
                        srsran_pdsch_res_t data[SRSRAN_MAX_CODEWORDS])
{
  if (!q || !sf_symbols || !data || !cfg || !sf || !channel) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  int tb_idx = -1, n_en = 0;
  for (int i = 0; i < SRSRAN_MAX_TB; i++) {
    if (cfg->grant.tb[i].enabled) {
      n_en++;
      tb_idx = tb_idx < 0 ? i : tb_idx;
    }
  }
  const bool dev = q->cell.nof_ports == 1 && q->nof_rx_antennas == 1 && cfg->grant.nof_tb == 1 && n_en == 1 && cfg->grant.nof_layers == 1 && !cfg->csi_enable &&
                   !cfg->meas_evm_en && (!cfg->power_scale || cfg->p_b == 0) && cfg->grant.tb[tb_idx].tbs > 0 && cfg->grant.tb[tb_idx].nof_bits && cfg->grant.nof_re &&
                   cfg->grant.tb[tb_idx].cw_idx == 0 && cfg->softbuffers.rx[tb_idx] && !data[tb_idx].crc && data[tb_idx].payload && !q->coworker_ptr &&
                   cfg->grant.nof_re <= q->max_re;
  if (!dev) {
    count(1, false);
    return srsran_pdsch_decode_ref(q, sf, cfg, channel, sf_symbols, data);
  }
  count(1, true);
  struct timeval t0;
  if (cfg->meas_time_en) {
    gettimeofday(&t0, NULL);
  }
  float scaling = 1.0f;
  if (cfg->power_scale) { /* :807-813 with rho_b = 1: nothing in the grid is touched, rho_a scales the equaliser */
    const float rho_a = srsran_convert_dB_to_amplitude(cfg->p_a);
    if (rho_a != 0.0f && isnormal(rho_a)) {
      scaling = rho_a;
    }
  }
  if (cfg->max_nof_iterations) {
    srsran_sch_set_max_noi(&q->dl_sch, cfg->max_nof_iterations);
  }
  const uint32_t lstart = SRSRAN_NOF_CTRL_SYMBOLS(q->cell, sf->cfi);
  if (srsran_pdsch_get(q, sf_symbols[0], q->symbols[0], &cfg->grant, lstart, sf->tti % 10) != (int)cfg->grant.nof_re ||
      srsran_pdsch_get(q, channel->ce[0][0], q->ce[0][0], &cfg->grant, lstart, sf->tti % 10) != (int)cfg->grant.nof_re) {
    ERROR("Error expecting %d symbols", cfg->grant.nof_re);
    return SRSRAN_ERROR;
  }
  const srsran_ra_tb_t* mcs = &cfg->grant.tb[tb_idx];
  srsran_hip_pdsch_rx_t g   = {.tb             = {.mod                = (uint32_t)mcs->mod,
                                                  .tbs                = (uint32_t)mcs->tbs,
                                                  .rv                 = (uint32_t)mcs->rv,
                                                  .nof_re             = cfg->grant.nof_re,
                                                  .seed               = srsran_hip_sequence_pdsch_seed(cfg->rnti, 0, 2 * (sf->tti % SRSRAN_NOF_SF_X_FRAME), q->cell.id),
                                                  .max_nof_iterations = q->dl_sch.max_iterations,
                                                  .llr_is_8bit        = q->llr_is_8bit,
                                                  .nl                 = 1},
                               .scaling        = scaling,
                               .noise_estimate = cfg->decoder_type == SRSRAN_MIMO_DECODER_ZF ? 0 : channel->noise_estimate};
  srsran_hip_grant_res_t r;
  /* q->d / q->e are where the reference leaves the equalised symbols and the descrambled soft bits, and callers do look (phy_dl_test.c:253-298, the
   * UE's constellation plots): CHAN_BIND_LEAN=1 passes NULL for both and saves the two copies */
  static int lean = -1;
  if (lean < 0) {
    lean = getenv("CHAN_BIND_LEAN") ? 1 : 0;
  }
  if (srsran_hip_pdsch_decode_dbg(&g, q->symbols[0], q->ce[0][0], cfg->softbuffers.rx[tb_idx], data[tb_idx].payload, &r, lean ? NULL : q->d[0], lean ? NULL : q->e[0]) !=
      SRSRAN_SUCCESS) {
    return SRSRAN_ERROR;
  }
  data[tb_idx].crc                  = r.crc_ok != 0;
  data[tb_idx].evm                  = NAN;
  data[tb_idx].avg_iterations_block = r.avg_iterations_block;
  q->dl_sch.avg_iterations          = r.avg_iterations_block;
  if (cfg->meas_time_en) {
    cfg->meas_time_value = elapsed_us(&t0);
  }
  return SRSRAN_SUCCESS;
}

/* ---- pdsch.c:1017.  Device path per codeword: CRC + segmentation + turbo coding + rate matching + scrambling + modulation in one call; layer mapping,
 * precoding and the resource mapping stay the reference's (srsran_layermap_type, srsran_precoding_type, srsran_pdsch_put). */
int srsran_pdsch_encode(srsran_pdsch_t* q, srsran_dl_sf_cfg_t* sf, srsran_pdsch_cfg_t* cfg, uint8_t* data[SRSRAN_MAX_CODEWORDS], cf_t* sf_symbols[SRSRAN_MAX_PORTS])
{
  if (!q || !cfg || !sf || !data || !sf_symbols) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  bool dev = q->nof_rx_antennas == 0 && cfg->grant.nof_tb >= 1 && cfg->grant.nof_re <= q->max_re && cfg->grant.nof_layers >= 1 &&
             cfg->grant.nof_layers <= SRSRAN_MAX_LAYERS && cfg->p_b < 4;
  for (uint32_t i = 0; i < q->cell.nof_ports && dev; i++) {
    dev = sf_symbols[i] != NULL;
  }
  for (int i = 0; i < SRSRAN_MAX_TB && dev; i++) {
    if (cfg->grant.tb[i].enabled) {
      dev = cfg->softbuffers.tx[i] && cfg->grant.tb[i].tbs > 0 && cfg->grant.tb[i].nof_bits == cfg->grant.nof_re * srsran_mod_bits_x_symbol(cfg->grant.tb[i].mod) &&
            cfg->grant.tb[i].cw_idx < SRSRAN_MAX_CODEWORDS;
    }
  }
  if (!dev) {
    count(2, false);
    return srsran_pdsch_encode_ref(q, sf, cfg, data, sf_symbols);
  }
  count(2, true);
  struct timeval t0;
  if (cfg->meas_time_en) {
    gettimeofday(&t0, NULL);
  }
  /* :486-492 (an eNB-side object has no receive antennas: the power allocation touches nothing and returns rho_a) */
  const float    rho_a   = srsran_convert_dB_to_amplitude(cfg->p_a) * ((q->cell.nof_ports == 1) ? 1.0f : M_SQRT2);
  const float    scaling = rho_a != 0.0f ? rho_a : 1.0f;
  const uint32_t nof_tb  = cfg->grant.nof_tb;
  const uint32_t nl      = cfg->grant.nof_layers != nof_tb ? 2 : 1; /* sch.c:632 */
  for (uint32_t tb_idx = 0; tb_idx < SRSRAN_MAX_TB; tb_idx++) {
    if (!cfg->grant.tb[tb_idx].enabled) {
      continue;
    }
    const srsran_ra_tb_t* mcs = &cfg->grant.tb[tb_idx];
    const uint32_t        cw  = mcs->cw_idx;
    srsran_hip_pdsch_tx_t g   = {.tb      = {.mod    = (uint32_t)mcs->mod,
                                             .tbs    = (uint32_t)mcs->tbs,
                                             .rv     = (uint32_t)mcs->rv,
                                             .nof_re = cfg->grant.nof_re,
                                             .seed   = srsran_hip_sequence_pdsch_seed(cfg->rnti, (int)cw, 2 * (sf->tti % SRSRAN_NOF_SF_X_FRAME), q->cell.id),
                                             .nl     = nl},
                                 .scaling = 1.0f};
    /* q->d[cw] <- the constellation points, q->e[cw] <- the scrambled coded bits, as pdsch.c:996-1012 leaves them */
    if (srsran_hip_pdsch_encode_dbg(&g, cfg->softbuffers.tx[tb_idx], data[tb_idx], q->d[cw], (uint8_t*)q->e[cw]) != SRSRAN_SUCCESS) {
      return SRSRAN_ERROR;
    }
  }
  if (q->cell.nof_ports == 1) { /* :1116-1121 */
    if (scaling == 1.0f) {
      memcpy(q->symbols[0], q->d[0], cfg->grant.nof_re * sizeof(cf_t));
    } else {
      srsran_vec_sc_prod_cfc(q->d[0], scaling, q->symbols[0], cfg->grant.nof_re);
    }
  }
  if (q->cell.nof_ports > 1) { /* :1087-1114 */
    cf_t* x[SRSRAN_MAX_LAYERS];
    int   nof_symbols;
    if (cfg->grant.nof_layers == nof_tb) {
      for (uint32_t i = 0; i < cfg->grant.nof_layers; i++) {
        x[i] = q->d[i];
      }
      nof_symbols = cfg->grant.nof_re;
    } else {
      for (uint32_t i = 0; i < cfg->grant.nof_layers; i++) {
        x[i] = q->x[i];
      }
      memset(&x[cfg->grant.nof_layers], 0, sizeof(cf_t*) * (SRSRAN_MAX_LAYERS - cfg->grant.nof_layers));
      nof_symbols = srsran_layermap_type(q->d, x, nof_tb, cfg->grant.nof_layers, (int[SRSRAN_MAX_CODEWORDS]){cfg->grant.nof_re, cfg->grant.nof_re}, cfg->grant.tx_scheme);
    }
    srsran_precoding_type(x, q->symbols, cfg->grant.nof_layers, q->cell.nof_ports, nof_tb == 1 ? cfg->grant.pmi : (cfg->grant.pmi + 1), nof_symbols, scaling,
                          cfg->grant.tx_scheme);
  }
  const uint32_t lstart = SRSRAN_NOF_CTRL_SYMBOLS(q->cell, sf->cfi);
  for (uint32_t i = 0; i < q->cell.nof_ports; i++) {
    srsran_pdsch_put(q, q->symbols[i], sf_symbols[i], &cfg->grant, lstart, sf->tti % 10);
  }
  if (cfg->meas_time_en) {
    cfg->meas_time_value = elapsed_us(&t0);
  }
  return SRSRAN_SUCCESS;
}

/* ---- sch.c:1194 (UE transmit side), without UCI: encode_tb + channel interleaver on the device */
int srsran_ulsch_encode(srsran_sch_t* q, srsran_pusch_cfg_t* cfg, uint8_t* data, srsran_uci_value_t* uci_data, uint8_t* g_bits, uint8_t* q_bits)
{
  if (!q || !cfg || !q_bits) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  const uint32_t Qm = srsran_mod_bits_x_symbol(cfg->grant.tb.mod);
  if (has_uci(cfg) || cfg->grant.tb.tbs <= 0 || !cfg->softbuffers.tx || Qm == 0 || cfg->grant.nof_symb == 0 || cfg->grant.tb.nof_bits % Qm ||
      (cfg->grant.tb.nof_bits / Qm) % cfg->grant.nof_symb) {
    count(3, false);
    return srsran_ulsch_encode_ref(q, cfg, data, uci_data, g_bits, q_bits);
  }
  count(3, true);
  srsran_cbsegm_t seg;
  if (srsran_cbsegm(&seg, (uint32_t)cfg->grant.tb.tbs)) {
    ERROR("Error computing segmentation for TBS=%d", cfg->grant.tb.tbs);
    return SRSRAN_ERROR;
  }
  cfg->K_segm              = seg.C1 * seg.K1 + seg.C2 * seg.K2;
  srsran_hip_grant_tb_t tb = {.mod = (uint32_t)cfg->grant.tb.mod, .tbs = (uint32_t)cfg->grant.tb.tbs, .rv = (uint32_t)cfg->grant.tb.rv,
                              .nof_re = cfg->grant.tb.nof_bits / Qm, .nl = 1};
  if (srsran_hip_ulsch_encode(&tb, cfg->grant.nof_symb, cfg->softbuffers.tx, data, q_bits) != SRSRAN_SUCCESS) {
    return SRSRAN_ERROR;
  }
  return 0; /* number of RI / ACK bits placed */
}
