/* tools/probe/warm_probe.c -- what the first grant of a process / of a worker thread costs, with and without the init-time warm-up.
 *   cc -O2 -I include tools/probe/warm_probe.c -o tools/probe/warm_probe -L srslte_amd/lib -lsrsran_phy_hip -Wl,-rpath,'$ORIGIN/../../srslte_amd/lib' -lpthread -lm
 *   tools/probe/warm_probe [cold|init|warmup N]
 * cold: no init hook at all (first call pays everything); init: srsran_rm_turbo_gentables() first, as srsran_sch_init does; warmup N: srsran_hip_warmup(N) first.
 * Prints microseconds of the 1st .. 4th PDSCH codeword decode (TBS 75376: 13 blocks of 5824, a size the warm-up did NOT use) and encode on the main
 * thread and on two fresh worker threads. */
#include <complex.h>
#include <pthread.h>
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

#define NCB 13
#define SB 18600
static void* worker(void* name)
{
  const uint32_t tbs = 75376, nof_re = 14400;
  int16_t*       rows[NCB];
  uint8_t *      keep[NCB], *txr[NCB];
  bool           flags[NCB];
  for (int i = 0; i < NCB; i++) {
    rows[i] = calloc(SB, 2);
    keep[i] = calloc(SB / 8, 1);
    txr[i]  = calloc(SB, 1);
  }
  srsran_softbuffer_rx_t rx = {NCB, SB, rows, keep, flags, false};
  srsran_softbuffer_tx_t tx = {NCB, SB, txr};
  float complex*         sym = calloc(nof_re, sizeof(float complex));
  uint8_t*               data = calloc(tbs / 8 + 64, 1);
  uint8_t*               out = calloc(tbs / 8 + 64, 1);
  for (uint32_t i = 0; i < tbs / 8; i++) {
    data[i] = (uint8_t)(i * 37 + 11);
  }
  srsran_hip_grant_tb_t tb = {SRSRAN_MOD_64QAM, tbs, 0, nof_re, 0x4d2c8a1, 10, 0, 1};
  srsran_hip_pdsch_tx_t gt = {tb, 1.0f};
  srsran_hip_pdsch_rx_t gr = {tb, 1.0f, 0.0f};
  double                te[4], td[4];
  int                   ok = 1;
  for (int k = 0; k < 4; k++) {
    double t0 = now_us();
    if (srsran_hip_pdsch_encode(&gt, &tx, data, (cf_t*)sym)) {
      ok = 0;
    }
    te[k] = now_us() - t0;
    for (int i = 0; i < NCB; i++) {
      memset(rows[i], 0, SB * 2);
      flags[i] = false;
    }
    srsran_hip_grant_res_t res;
    t0 = now_us();
    if (srsran_hip_pdsch_decode(&gr, (cf_t*)sym, NULL, &rx, out, &res) || !res.crc_ok || memcmp(out, data, tbs / 8)) {
      ok = 0;
    }
    td[k] = now_us() - t0;
  }
  printf("%-8s %s  encode us: %8.0f %6.0f %6.0f %6.0f   decode us: %8.0f %6.0f %6.0f %6.0f\n", (const char*)name, ok ? "ok " : "BAD", te[0], te[1], te[2], te[3],
         td[0], td[1], td[2], td[3]);
  return NULL;
}

int main(int argc, char** argv)
{
  const char* mode = argc > 1 ? argv[1] : "cold";
  double      t0   = now_us();
  if (!strcmp(mode, "init")) {
    srsran_rm_turbo_gentables();
  } else if (!strcmp(mode, "warmup")) {
    srsran_hip_warmup(argc > 2 ? (uint32_t)atoi(argv[2]) : 1);
  }
  printf("mode %s: init-time work %.1f ms\n", mode, (now_us() - t0) * 1e-3);
  worker("main");
  pthread_t th;
  pthread_create(&th, NULL, worker, "worker1");
  pthread_join(th, NULL);
  pthread_create(&th, NULL, worker, "worker2");
  pthread_join(th, NULL);
  return 0;
}
