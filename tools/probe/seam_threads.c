/* seam_threads.c -- the transport-block seam under the reference's threading model, measured from C (a Python harness cannot: its threads
 * share the interpreter lock and every return from the library waits for it).  N worker threads, each with its own srsran_sch_t head, soft
 * buffer and buffers, decode one transport block per call through srsran_hip_decode_tb_cb (= decode_tb_cb, sch.c:370) at the same time;
 * the block is made by the library's own transmit entry (srsran_hip_encode_tb), noise-free: one half iteration per code block.
 * Build: gcc -O2 -I../../include seam_threads.c -o seam_threads -L../../srslte_amd/lib -lsrsran_phy_hip -Wl,-rpath,'$ORIGIN/../../srslte_amd/lib' -lpthread -lm */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "srsran_amd/phy_sch_abi.h"

#define TBS 75376
#define QM 6
#define G 100800
#define CALLS 400
#define SB 18600

static int16_t           e_bits[G];
static uint8_t           payload[TBS / 8 + 8];
static pthread_barrier_t bar;

static double now_us(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

typedef struct {
  double lat[CALLS];
  int    ok;
} result_t;

static void* worker(void* arg)
{
  result_t*              r = arg;
  srsran_cbsegm_t        cs;
  srsran_softbuffer_rx_t sb;
  srsran_hip_sch_head_t  q = {10, 0.f, false};
  srsran_cbsegm(&cs, TBS);
  sb.max_cb = cs.C, sb.max_cb_size = SB;
  sb.buffer_f = calloc(cs.C, sizeof(int16_t*)), sb.data = calloc(cs.C, sizeof(uint8_t*)), sb.cb_crc = calloc(cs.C, sizeof(bool));
  for (uint32_t i = 0; i < cs.C; i++) {
    sb.buffer_f[i] = calloc(SB, sizeof(int16_t)), sb.data[i] = calloc(SB / 8, 1);
  }
  uint8_t* data = calloc(TBS / 8 + 16, 1);
  int16_t* e    = malloc(sizeof(e_bits));
  memcpy(e, e_bits, sizeof(e_bits));
  r->ok = 0;
  pthread_barrier_wait(&bar);
  for (int i = 0; i < CALLS + 20; i++) {
    memset(sb.cb_crc, 0, cs.C); /* srsran_softbuffer_rx_reset: the rows are still zero (decoded blocks' rows are not written back) */
    const double t0   = now_us();
    const bool   good = srsran_hip_decode_tb_cb(&q, &sb, &cs, QM, 0, G, e, data);
    const double dt   = now_us() - t0;
    if (i >= 20) {
      r->lat[i - 20] = dt;
      r->ok += good && !memcmp(data, payload, TBS / 8);
    }
  }
  return NULL;
}

static int cmp(const void* a, const void* b)
{
  return *(const double*)a < *(const double*)b ? -1 : 1;
}

int main(void)
{
  /* the block: random payload -> srsran_hip_encode_tb -> +-100 soft bits */
  srsran_cbsegm_t        cs;
  srsran_softbuffer_tx_t tx;
  srsran_cbsegm(&cs, TBS);
  tx.max_cb = cs.C, tx.max_cb_size = SB, tx.buffer_b = calloc(cs.C, sizeof(uint8_t*));
  for (uint32_t i = 0; i < cs.C; i++) {
    tx.buffer_b[i] = calloc(SB, 1);
  }
  srand(7);
  for (int i = 0; i < TBS / 8; i++) {
    payload[i] = (uint8_t)rand();
  }
  uint8_t* packed = calloc(G / 8 + 8, 1);
  if (srsran_hip_encode_tb(&tx, &cs, QM, 0, G, payload, packed)) {
    fprintf(stderr, "encode failed (no HIP device?)\n");
    return 1;
  }
  for (int i = 0; i < G; i++) {
    e_bits[i] = ((packed[i >> 3] >> (7 - (i & 7))) & 1) ? 100 : -100; /* the reference's sign convention: bit 1 <-> positive soft bit */
  }
  static const int counts[] = {1, 2, 3, 4, 8};
  for (unsigned c = 0; c < sizeof(counts) / sizeof(counts[0]); c++) {
    const int n = counts[c];
    pthread_t th[8];
    result_t* res = calloc(n, sizeof(result_t));
    pthread_barrier_init(&bar, NULL, n);
    const double t0 = now_us();
    for (int t = 0; t < n; t++) {
      pthread_create(&th[t], NULL, worker, &res[t]);
    }
    for (int t = 0; t < n; t++) {
      pthread_join(th[t], NULL);
    }
    const double wall = now_us() - t0;
    double*      all  = malloc(sizeof(double) * n * CALLS);
    int          ok   = 0;
    for (int t = 0; t < n; t++) {
      memcpy(all + t * CALLS, res[t].lat, sizeof(double) * CALLS);
      ok += res[t].ok;
    }
    qsort(all, n * CALLS, sizeof(double), cmp);
    printf("%d worker thread(s): p50 %.1f us, p99 %.1f us per transport block (13 code blocks, one half iteration each), %d of %d decoded, %.0f blocks/s = %.0f Mbit/s in all\n", n,
           all[n * CALLS / 2], all[(int)(n * CALLS * 0.99) - 1], ok, n * CALLS, n * (CALLS + 20) / (wall * 1e-6), n * (CALLS + 20) / (wall * 1e-6) * TBS / 1e6);
    free(all), free(res);
  }
  return 0;
}
