/*
 * c_caller.c -- TEST INFRASTRUCTURE (our own code): a plain C program on top of the library's C ABI (the headers of include/srsran_amd compiled as C).
 *
 * Checks the one entry point that returns a complex number BY VALUE (srsran_cp_synch_corr_output, cp.h:46): the library is C++
 * (std::complex<float>), the caller is C (float _Complex).  The correlation of srsran_cp_synch (cp.c:60-79) is recomputed here in double and
 * compared with what the call returns, for every offset.  Exit code 0 = all within 1e-4.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "srsran_amd/phy_abi.h"
#include "srsran_amd/phy_sync_abi.h"

int main(void)
{
  const uint32_t N = 128, cp_len = 9, nof_symbols = 6, max_offset = 40;
  const uint32_t len = max_offset + nof_symbols * (N + cp_len + 1) + N;
  cf_t*          x   = (cf_t*)calloc(len, sizeof(cf_t));
  uint32_t       s   = 12345u;
  for (uint32_t i = 0; i < len; i++) {
    s        = s * 1664525u + 1013904223u;
    float re = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    s        = s * 1664525u + 1013904223u;
    float im = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    x[i]     = re + im * _Complex_I;
  }
  /* a cyclic prefix structure starting at sample 17: the tail of every symbol repeated in front of it (the first symbol of 7 has one more, cp.c:68) */
  for (uint32_t sym = 0, pos = 17; sym < nof_symbols; sym++) {
    const uint32_t cplen = (sym % 7) ? cp_len : cp_len + 1;
    cf_t*          p     = x + pos;
    for (uint32_t i = 0; i < cplen; i++) {
      p[i] = p[i + N];
    }
    pos += N + cplen;
  }
  srsran_cp_synch_t q;
  if (srsran_cp_synch_init(&q, N)) {
    fprintf(stderr, "srsran_cp_synch_init failed\n");
    return 2;
  }
  const uint32_t peak = srsran_cp_synch(&q, x, max_offset, nof_symbols, cp_len);
  int            bad  = 0;
  for (uint32_t off = 0; off < max_offset; off++) {
    double re = 0, im = 0; /* cp.c:60-79: sum over symbols of conj(in[.. + N]) . in[..] over cp_len samples, / nof_symbols */
    for (uint32_t sym = 0, pos = off; sym < nof_symbols; sym++) {
      const uint32_t cplen = (sym % 7) ? cp_len : cp_len + 1;
      const cf_t*    p     = x + pos;
      pos += N + cplen;
      for (uint32_t i = 0; i < cplen; i++) {
        const double ar = crealf(p[i]), ai = cimagf(p[i]), br = crealf(p[i + N]), bi = cimagf(p[i + N]);
        re += ar * br + ai * bi;
        im += ai * br - ar * bi;
      }
    }
    re /= nof_symbols;
    im /= nof_symbols;
    const cf_t got = srsran_cp_synch_corr_output(&q, off);
    if (fabs(crealf(got) - re) > 1e-4 || fabs(cimagf(got) - im) > 1e-4) {
      if (bad++ < 4) {
        fprintf(stderr, "offset %u: got %g%+gi, expected %g%+gi\n", off, crealf(got), cimagf(got), re, im);
      }
    }
  }
  printf("peak at %u (expected 17), %d of %u values off\n", peak, bad, max_offset);
  srsran_cp_synch_free(&q);
  free(x);
  return (bad == 0 && peak == 17) ? 0 : 1;
}
