/* tools/probe/tti_probe.c -- the PUSCH grants of one TTI through the grant-level entry points, from C (host grids in, payload out):
 *   (a) one srsran_hip_pusch_decode per grant, one after the other (what a worker that loops over its UEs does, cc_worker.cc:359-371)
 *   (b) all of them in ONE srsran_hip_pusch_decode_multi
 * for a few splits of a 100-PRB cell, plus the single-grant time per allocation size.  The signal is made by the library's own transmit side
 * (srsran_hip_ulsch_encode -> srsran_hip_modulate_bytes with the PUSCH seed -> srsran_dft_precoding -> grid); every payload is checked.
 *   cc -O2 -I include tools/probe/tti_probe.c -o tools/probe/tti_probe -L srslte_amd/lib -lsrsran_phy_hip -Wl,-rpath,'$ORIGIN/../../srslte_amd/lib' -lm */
#include <complex.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "srsran_amd/phy_chan_abi.h"

static double now_us(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

#define SB 18600
#define NPRB 100
#define NRE (14 * 12 * NPRB)

typedef struct {
  srsran_hip_pusch_rx_t  g;
  float complex*         grid;
  float complex*         ce;
  uint8_t*               payload;
  uint8_t*               out;
  srsran_softbuffer_rx_t rx;
  int16_t*               rows[13];
  uint8_t*               keep[13];
  bool                   flags[13];
} ue_t;

/* largest transport block (bits, no filler) that n_cb code blocks of size K carry */
static uint32_t tbs_of(uint32_t K, uint32_t n_cb)
{
  return n_cb > 1 ? n_cb * (K - 24) - 24 : K - 24;
}

static void make_ue(ue_t* u, uint32_t first_prb, uint32_t L, uint32_t mod, uint32_t tbs, uint32_t id, srsran_dft_precoding_t* pre)
{
  const uint32_t nsymb = 12, nsc = 12 * L, nof_re = nsymb * nsc, qm = 2 * mod;
  memset(u, 0, sizeof(*u));
  u->grid    = calloc(NRE, sizeof(float complex));
  u->ce      = calloc(NRE, sizeof(float complex));
  u->payload = calloc(tbs / 8 + 64, 1);
  u->out     = calloc(tbs / 8 + 64, 1);
  for (uint32_t i = 0; i < tbs / 8; i++) {
    u->payload[i] = (uint8_t)(i * 131 + id * 17 + 5);
  }
  for (int i = 0; i < 13; i++) {
    u->rows[i] = calloc(SB, 2);
    u->keep[i] = calloc(SB / 8, 1);
  }
  u->rx = (srsran_softbuffer_rx_t){13, SB, u->rows, u->keep, u->flags, false};
  for (uint32_t i = 0; i < NRE; i++) {
    u->ce[i] = 1.0f;
  }
  srsran_hip_grant_tb_t tb = {mod, tbs, 0, nof_re, srsran_hip_sequence_pusch_seed((uint16_t)(0x100 + id), 4, 77), 10, 0, 1};
  u->g = (srsran_hip_pusch_rx_t){tb, NPRB, 7, {first_prb, first_prb}, L, 0, 0.0f, 0};
  /* transmit side: encode + interleave, scramble + modulate, transform precoding, resource mapping */
  uint8_t*       txr[13];
  for (int i = 0; i < 13; i++) {
    txr[i] = calloc(SB, 1);
  }
  srsran_softbuffer_tx_t tx = {13, SB, txr};
  uint8_t*               q  = calloc(nof_re * qm / 8 + 8, 1);
  float complex*         d  = calloc(nof_re, sizeof(float complex));
  float complex*         z  = calloc(nof_re, sizeof(float complex));
  if (srsran_hip_ulsch_encode(&tb, nsymb, &tx, u->payload, q) || srsran_hip_modulate_bytes(mod, q, (cf_t*)d, nof_re * qm, tb.seed, 1, 1.0f) != (int)nof_re ||
      srsran_dft_precoding(pre, (cf_t*)d, (cf_t*)z, L, nsymb)) {
    fprintf(stderr, "transmit side failed\n");
    exit(1);
  }
  uint32_t row = 0;
  for (uint32_t sym = 0; sym < 14; sym++) {
    if (sym == 3 || sym == 10) {
      continue;
    }
    memcpy(&u->grid[(sym * NPRB + first_prb) * 12], &z[row * nsc], nsc * sizeof(float complex));
    row++;
  }
  free(q), free(d), free(z);
  for (int i = 0; i < 13; i++) {
    free(txr[i]);
  }
}

static void reset(ue_t* u)
{
  for (int i = 0; i < 13; i++) {
    memset(u->rows[i], 0, SB * 2);
    u->flags[i] = false;
  }
}

static int cmp_d(const void* a, const void* b)
{
  return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b);
}

/* TTI_PROBE_ONLY=<split index>,<mode: 1 single, 2 loop, 4 multi; sum> restricts the run (for a kernel timeline of one kind of call) */
int main(void)
{
  int only_split = -1, modes = 7;
  if (getenv("TTI_PROBE_ONLY")) {
    sscanf(getenv("TTI_PROBE_ONLY"), "%d,%d", &only_split, &modes);
  }
  srsran_hip_warmup(1);
  srsran_dft_precoding_t pre;
  if (srsran_dft_precoding_init_tx(&pre, NPRB)) {
    return 1;
  }
  /* splits of the 100 PRB: (number of UEs, PRBs each, modulation, transport block) */
  const struct {
    uint32_t n, L, mod, K, ncb;
    const char* what;
  } splits[] = {{1, 100, 3, 6144, 12, "1 UE x 100 PRB, 64-QAM, 12 code blocks of 6144"},
                {2, 50, 3, 6144, 6, "2 UEs x 50 PRB, 64-QAM, 6 code blocks each"},
                {4, 25, 3, 6144, 3, "4 UEs x 25 PRB, 64-QAM, 3 code blocks each"},
                {8, 12, 2, 5504, 1, "8 UEs x 12 PRB, 16-QAM, one code block of 5504"},
                {16, 6, 2, 2816, 1, "16 UEs x 6 PRB, 16-QAM, one code block of 2816"},
                {25, 4, 1, 1024, 1, "25 UEs x 4 PRB, QPSK, one code block of 1024"},
                {25, 4, 1, 352, 1, "25 UEs x 4 PRB, QPSK, one block of 352 (scalar decoder)"},
                {50, 2, 1, 176, 1, "50 UEs x 2 PRB, QPSK, one block of 176 (scalar decoder)"}};
  printf("%-52s %12s %12s %12s\n", "split of one 100-PRB subframe", "single us", "loop us", "multi us");
  for (unsigned s = 0; s < sizeof(splits) / sizeof(splits[0]); s++) {
    if (only_split >= 0 && (int)s != only_split) {
      continue;
    }
    const uint32_t n   = splits[s].n;
    ue_t*          ue  = calloc(n, sizeof(ue_t));
    const uint32_t tbs = tbs_of(splits[s].K, splits[s].ncb);
    for (uint32_t i = 0; i < n; i++) {
      make_ue(&ue[i], i * splits[s].L, splits[s].L, splits[s].mod, tbs, i + 100 * s, &pre);
    }
    srsran_hip_pusch_rx_t*   g   = calloc(n, sizeof(*g));
    const cf_t**             gr  = calloc(n, sizeof(*gr));
    const cf_t**             ce  = calloc(n, sizeof(*ce));
    srsran_softbuffer_rx_t** sb  = calloc(n, sizeof(*sb));
    uint8_t**                out = calloc(n, sizeof(*out));
    srsran_hip_grant_res_t*  res = calloc(n, sizeof(*res));
    for (uint32_t i = 0; i < n; i++) {
      g[i] = ue[i].g, gr[i] = (const cf_t*)ue[i].grid, ce[i] = (const cf_t*)ue[i].ce, sb[i] = &ue[i].rx, out[i] = ue[i].out;
    }
    enum { REP = 60 };
    double t1[REP], tl[REP], tm[REP];
    int    ok = 1;
    for (int r = 0; r < REP; r++) {
      reset(&ue[0]);
      double t0 = now_us();
      t1[r] = tl[r] = tm[r] = 0;
      if (modes & 1) {
        ok &= srsran_hip_pusch_decode(&g[0], gr[0], ce[0], sb[0], out[0], &res[0]) == 0 && res[0].crc_ok;
        t1[r] = now_us() - t0;
      }
      for (uint32_t i = 0; i < n; i++) {
        reset(&ue[i]);
      }
      if (modes & 2) {
        t0 = now_us();
        for (uint32_t i = 0; i < n; i++) {
          ok &= srsran_hip_pusch_decode(&g[i], gr[i], ce[i], sb[i], out[i], &res[i]) == 0 && res[i].crc_ok;
        }
        tl[r] = now_us() - t0;
        for (uint32_t i = 0; i < n; i++) {
          ok &= memcmp(ue[i].out, ue[i].payload, tbs / 8) == 0;
          memset(ue[i].out, 0, tbs / 8);
          reset(&ue[i]);
        }
      }
      if (modes & 4) {
        t0 = now_us();
        ok &= srsran_hip_pusch_decode_multi(n, g, gr, ce, sb, out, res) == 0;
        tm[r] = now_us() - t0;
        for (uint32_t i = 0; i < n; i++) {
          ok &= res[i].crc_ok && memcmp(ue[i].out, ue[i].payload, tbs / 8) == 0;
        }
      }
    }
    qsort(t1, REP, sizeof(double), cmp_d), qsort(tl, REP, sizeof(double), cmp_d), qsort(tm, REP, sizeof(double), cmp_d);
    printf("%-52s %12.1f %12.1f %12.1f  %s (TBS %u per UE, %.2f decoder iterations per block)\n", splits[s].what, t1[REP / 2], tl[REP / 2], tm[REP / 2],
           ok ? "ok" : "BAD", tbs, res[0].avg_iterations_block);
  }
  /* ---- downlink: the PDSCH codewords of one TTI, one srsran_hip_pdsch_encode per UE against one srsran_hip_pdsch_encode_multi */
  if (only_split < 0) {
    const struct {
      uint32_t n, prb, mod, tbs;
      const char* what;
    } dl[] = {{1, 100, 3, 75376, "1 UE x 100 PRB, 64-QAM (13 code blocks)"},
              {4, 25, 3, 18336, "4 UEs x 25 PRB, 64-QAM (3 code blocks each)"},
              {8, 12, 2, 5736, "8 UEs x 12 PRB, 16-QAM (one code block)"},
              {16, 6, 2, 2792, "16 UEs x 6 PRB, 16-QAM"},
              {25, 4, 1, 1000, "25 UEs x 4 PRB, QPSK"}};
    printf("%-52s %12s %12s\n", "PDSCH codewords of one TTI (encode)", "loop us", "multi us");
    for (unsigned s = 0; s < sizeof(dl) / sizeof(dl[0]); s++) {
      const uint32_t n = dl[s].n, nof_re = dl[s].prb * 12 * 11, tbs = dl[s].tbs; /* (11 of 14 symbols carry PDSCH, reference signals not taken out) */
      srsran_hip_pdsch_tx_t*   g   = calloc(n, sizeof(*g));
      srsran_softbuffer_tx_t*  sb  = calloc(n, sizeof(*sb));
      srsran_softbuffer_tx_t** sbp = calloc(n, sizeof(*sbp));
      uint8_t**                pay = calloc(n, sizeof(*pay));
      cf_t **                  o1 = calloc(n, sizeof(*o1)), **o2 = calloc(n, sizeof(*o2));
      for (uint32_t i = 0; i < n; i++) {
        g[i] = (srsran_hip_pdsch_tx_t){{dl[s].mod, tbs, 0, nof_re, 0x1234u + i, 0, 0, 1}, 1.0f};
        uint8_t** rows = calloc(13, sizeof(uint8_t*));
        for (int r = 0; r < 13; r++) {
          rows[r] = calloc(SB, 1);
        }
        sb[i]  = (srsran_softbuffer_tx_t){13, SB, rows};
        sbp[i] = &sb[i];
        pay[i] = calloc(tbs / 8 + 8, 1);
        for (uint32_t k = 0; k < tbs / 8; k++) {
          pay[i][k] = (uint8_t)(k * 37 + i * 11 + 3);
        }
        o1[i] = calloc(nof_re, sizeof(cf_t)), o2[i] = calloc(nof_re, sizeof(cf_t));
      }
      enum { REP = 60 };
      double tl[REP], tm[REP];
      int    ok = 1;
      for (int r = 0; r < REP; r++) {
        double t0 = now_us();
        for (uint32_t i = 0; i < n; i++) {
          ok &= srsran_hip_pdsch_encode(&g[i], &sb[i], pay[i], o1[i]) == 0;
        }
        tl[r] = now_us() - t0;
        t0    = now_us();
        ok &= srsran_hip_pdsch_encode_multi(n, g, sbp, pay, o2) == 0;
        tm[r] = now_us() - t0;
        for (uint32_t i = 0; i < n; i++) {
          ok &= memcmp(o1[i], o2[i], nof_re * sizeof(cf_t)) == 0;
        }
      }
      qsort(tl, REP, sizeof(double), cmp_d), qsort(tm, REP, sizeof(double), cmp_d);
      printf("%-52s %12.1f %12.1f  %s (TBS %u per UE)\n", dl[s].what, tl[REP / 2], tm[REP / 2], ok ? "ok" : "BAD", tbs);
    }
  }
  return 0;
}
