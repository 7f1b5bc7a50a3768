/*
 * orc_ofdm.c -- CPU restatement of the reference OFDM (de)modulator and DFT wrapper
 * (TEST INFRASTRUCTURE ONLY).
 *
 * Restates:
 *   lib/src/phy/dft/ofdm.c:38-212   (geometry, window offset, dc rule)
 *   lib/src/phy/dft/ofdm.c:334-362  (frequency-shift table)
 *   lib/src/phy/dft/ofdm.c:387-422,453-466 (rx), :487-536,562-576 (tx)
 *   lib/src/phy/dft/dft_fftw.c:297-354 (mirror / dc / norm options of srsran_dft_run_c)
 *   lib/src/phy/common/phy_common.c:342-385, phy_common.h:125 (symbol size, CP length)
 *
 * The reference delegates the transform itself to FFTW3f (system library, absent from this image, so
 * the reference's OFDM path is unbuildable here).  The oracle evaluates the DFT in float64 with an
 * exact-twiddle mixed-radix recursion and rounds once at the end: it is MORE accurate than FFTW3f's
 * float32 output, and the reference's own acceptance threshold (ofdm_test.c:176, 1e-4) is what the
 * parity tests use.
 */
#include "oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cd;

int orc_symbol_sz_power2(uint32_t nof_prb)
{
  if (nof_prb <= 6) {
    return 128;
  } else if (nof_prb <= 15) {
    return 256;
  } else if (nof_prb <= 25) {
    return 512;
  } else if (nof_prb <= 50) {
    return 1024;
  } else if (nof_prb <= 75) {
    return 1536;
  } else if (nof_prb <= 110) {
    return 2048;
  }
  return -1;
}

int orc_symbol_sz(uint32_t nof_prb)
{
  if (nof_prb == 0) {
    return -1;
  }
  if (nof_prb <= 6) {
    return 128;
  } else if (nof_prb <= 15) {
    return 256;
  } else if (nof_prb <= 25) {
    return 384;
  } else if (nof_prb <= 50) {
    return 768;
  } else if (nof_prb <= 75) {
    return 1024;
  } else if (nof_prb <= 110) {
    return 1536;
  }
  return -1;
}

int orc_cp_len(uint32_t symbol_sz, int c)
{
  return (int)ceilf((((float)(c) * (symbol_sz)) / 2048.0f));
}

/* ---- float64 DFT: recursive decimation in time over the smallest prime factor, O(N^2) for primes.
 * sign = -1 forward, +1 backward.  Twiddles come from one table of exp(sign*2*pi*i*k/n_root). */
static void dft_rec(const cd* in, int stride, cd* out, int n, const cd* tw, int tw_stride)
{
  if (n == 1) {
    out[0] = in[0];
    return;
  }
  int p = 0;
  for (int f = 2; f * f <= n; f++) {
    if (n % f == 0) {
      p = f;
      break;
    }
  }
  if (!p) { /* prime length: direct sum */
    for (int k = 0; k < n; k++) {
      cd acc = 0;
      for (int j = 0; j < n; j++) {
        acc += in[(size_t)j * stride] * tw[(size_t)((long)j * k % n) * tw_stride];
      }
      out[k] = acc;
    }
    return;
  }
  int m   = n / p;
  cd* sub = malloc(sizeof(cd) * n);
  for (int r = 0; r < p; r++) {
    dft_rec(in + (size_t)r * stride, stride * p, sub + (size_t)r * m, m, tw, tw_stride * p);
  }
  for (int k = 0; k < n; k++) {
    cd acc = 0;
    for (int r = 0; r < p; r++) {
      acc += sub[(size_t)r * m + (k % m)] * tw[(size_t)((long)r * k % n) * tw_stride];
    }
    out[k] = acc;
  }
  free(sub);
}

static void dft_f64(const cd* in, cd* out, int n, int backward)
{
  cd*    tw = malloc(sizeof(cd) * n);
  double sg = backward ? 1.0 : -1.0;
  for (int k = 0; k < n; k++) {
    tw[k] = cexp(I * sg * 2.0 * M_PI * (double)k / (double)n);
  }
  dft_rec(in, 1, out, n, tw, 1);
  free(tw);
}

void orc_dft_c(const float* in, float* out, int n, int backward, int mirror, int dc, int norm)
{
  cd* a = calloc(n, sizeof(cd));
  cd* b = calloc(n, sizeof(cd));
  int offset = dc ? 1 : 0;
  /* copy_pre, dft_fftw.c:297-308 */
  if (mirror && backward) {
    int hlen = n / 2;
    for (int i = 0; i < n - hlen - offset; i++) {
      a[offset + i] = in[2 * (hlen + i)] + I * in[2 * (hlen + i) + 1];
    }
    for (int i = 0; i < hlen; i++) {
      a[n - hlen + i] = in[2 * i] + I * in[2 * i + 1];
    }
  } else {
    for (int i = 0; i < n; i++) {
      a[i] = in[2 * i] + I * in[2 * i + 1];
    }
  }
  dft_f64(a, b, n, backward);
  if (norm) {
    double s = 1.0 / sqrt((double)n);
    for (int i = 0; i < n; i++) {
      b[i] *= s;
    }
  }
  /* copy_post, dft_fftw.c:310-320 */
  if (mirror && !backward) {
    int hlen = (n + 1) / 2;
    int k    = 0;
    for (int i = hlen; i < n; i++, k++) {
      out[2 * k]     = (float)creal(b[i]);
      out[2 * k + 1] = (float)cimag(b[i]);
    }
    for (int i = offset; i < hlen; i++, k++) {
      out[2 * k]     = (float)creal(b[i]);
      out[2 * k + 1] = (float)cimag(b[i]);
    }
  } else {
    for (int i = 0; i < n; i++) {
      out[2 * i]     = (float)creal(b[i]);
      out[2 * i + 1] = (float)cimag(b[i]);
    }
  }
  free(a);
  free(b);
}

typedef struct {
  int    N, nsym, cp0, cp1, nof_re, slot_sz, sf_sz, dc, win_n, shift_on;
  int    mbsfn;      /* slot 0 is laid out by ofdm_rx_slot_mbsfn / ofdm_tx_slot_mbsfn */
  int    mpos[6];    /* start of the useful part of each of its 6 symbols */
  int    mcp[6];     /* and their cyclic prefix lengths */
  double norm;
} geom_t;

/* symbol positions of the MBSFN slot.  rx: ofdm.c:424-437, tx: ofdm.c:538-555 (same layout for region 1, 2) */
static int mbsfn_layout(geom_t* g, int region, int tx)
{
  int N = g->N, ext = orc_cp_len(N, 512), n0 = orc_cp_len(N, 160), n1 = orc_cp_len(N, 144);
  int guard = region == 1 ? ext - n0 : 2 * ext - n0 - n1; /* SRSRAN_NON_MBSFN_REGION_GUARD_LENGTH, phy_common.h:166 */
  int at = 0;
  for (int i = 0; i < 6; i++) {
    if (!tx) {
      if (i == region) {
        at += guard;
      }
      g->mcp[i] = i >= region ? ext : (i == 0 ? n0 : n1);
      at += g->mcp[i];
      g->mpos[i] = at;
      at += N;
    } else {
      g->mcp[i]  = i > region - 1 ? ext : (i == 0 ? n0 : n1);
      g->mpos[i] = at + g->mcp[i];
      at += N + g->mcp[i];
      if (i == region - 1) {
        at += guard;
      }
    }
    if (g->mpos[i] + N > g->sf_sz) {
      return -1;
    }
  }
  return 0;
}

static int geometry(const orc_ofdm_cfg_t* cfg, geom_t* g)
{
  int N = cfg->symbol_sz ? (int)cfg->symbol_sz : orc_symbol_sz(cfg->nof_prb);
  if (N <= 0) {
    return -1;
  }
  g->N        = N;
  g->nsym     = cfg->cp_ext ? 6 : 7;
  g->cp0      = cfg->cp_ext ? orc_cp_len(N, 512) : orc_cp_len(N, 160);
  g->cp1      = cfg->cp_ext ? orc_cp_len(N, 512) : orc_cp_len(N, 144);
  g->nof_re   = 12 * (int)cfg->nof_prb;
  g->slot_sz  = N * 15 / 2;
  g->sf_sz    = N * 15;
  g->shift_on = isnormal(cfg->freq_shift_f);
  g->dc       = (!cfg->keep_dc && !g->shift_on) ? 1 : 0; /* ofdm.c:209 */
  g->win_n    = 0;
  if (isnormal(cfg->rx_window_offset)) { /* ofdm.c:126-131 */
    float off = cfg->rx_window_offset;
    off       = off < 0 ? 0 : off;
    off       = off > 100 ? 100 : off;
    g->win_n  = (int)roundf((float)g->cp1 * off);
  }
  g->norm  = cfg->normalize ? (double)(1.0f / sqrtf((float)N)) : 0.0;
  g->mbsfn = 0;
  return 0;
}

/* ofdm.c:344-356: the reference builds this table in float (cexpf of a float-rounded phase) */
static void shift_table(const geom_t* g, float freq_shift, float _Complex* tab)
{
  float _Complex* ptr = tab;
  for (int n = 0; n < 2; n++) {
    for (int i = 0; i < g->nsym; i++) {
      int cplen = i == 0 ? g->cp0 : g->cp1;
      for (int t = 0; t < g->N + cplen; t++) {
        ptr[t] = cexpf(I * 2 * M_PI * ((float)t - (float)cplen) * freq_shift / g->N);
      }
      ptr += g->N + cplen;
    }
  }
}

int orc_ofdm_rx_sf(const orc_ofdm_cfg_t* cfg, const float* in, float* out)
{
  geom_t g;
  if (geometry(cfg, &g) || g.nof_re > g.N - g.dc) {
    return -1;
  }
  if (cfg->mbsfn_region) { /* ofdm.c:459-463: slot 0 = ofdm_rx_slot_mbsfn, slot 1 = the regular path */
    if (!cfg->cp_ext || mbsfn_layout(&g, cfg->mbsfn_region, 0)) {
      return -1;
    }
    g.mbsfn = 1;
  }
  const int       N = g.N;
  float _Complex* sh = NULL;
  if (g.shift_on) {
    sh = malloc(sizeof(float _Complex) * g.sf_sz);
    shift_table(&g, cfg->freq_shift_f, sh);
  }
  cd* x = malloc(sizeof(cd) * N);
  cd* X = malloc(sizeof(cd) * N);
  for (int slot = 0; slot < 2; slot++) {
    for (int l = 0; l < g.nsym; l++) {
      const int special = g.mbsfn && slot == 0; /* no window offset there: plain srsran_dft_run_c, ofdm.c:432 */
      const int win_n   = special ? 0 : g.win_n;
      int pos = special ? g.mpos[l] : slot * g.slot_sz + g.cp0 + l * (N + g.cp1) - g.win_n; /* ofdm.c:157-166 */
      for (int n = 0; n < N; n++) {
        float _Complex s = in[2 * (pos + n)] + I * in[2 * (pos + n) + 1];
        if (sh) {
          s = s * sh[pos + n]; /* float product, ofdm.c:455-457 */
        }
        x[n] = s;
      }
      dft_f64(x, X, N, 0);
      float* o = out + 2 * (size_t)(slot * g.nsym + l) * g.nof_re;
      for (int k = 0; k < g.nof_re; k++) {
        int f = k < g.nof_re / 2 ? N - g.nof_re / 2 + k : g.dc + k - g.nof_re / 2; /* ofdm.c:410-411 */
        cd  v = X[f];
        if (win_n) { /* ofdm.c:134-136,405-407 */
          float _Complex r = cexpf(I * M_PI * 2.0f * (float)win_n * (float)f / (float)N);
          v *= (cd)r;
        }
        if (g.norm != 0.0) {
          v *= g.norm;
        }
        o[2 * k]     = (float)creal(v);
        o[2 * k + 1] = (float)cimag(v);
      }
    }
  }
  free(x);
  free(X);
  free(sh);
  return 0;
}

int orc_ofdm_tx_sf(const orc_ofdm_cfg_t* cfg, const float* in, float* out)
{
  geom_t g;
  if (geometry(cfg, &g) || g.nof_re > g.N - g.dc) {
    return -1;
  }
  if (cfg->mbsfn_region) { /* ofdm.c:569-571.  Guard carriers of the MBSFN slot are zero here, as on the first
                            * call of the reference (later calls leak stale scratch of slot 1 into them) */
    if (!cfg->cp_ext || mbsfn_layout(&g, cfg->mbsfn_region, 1)) {
      return -1;
    }
    g.mbsfn = 1;
  }
  const int       N = g.N;
  float _Complex* sh = NULL;
  if (g.shift_on) {
    sh = malloc(sizeof(float _Complex) * g.sf_sz);
    shift_table(&g, cfg->freq_shift_f, sh);
  }
  cd* X = malloc(sizeof(cd) * N);
  cd* x = malloc(sizeof(cd) * N);
  memset(out, 0, sizeof(float) * 2 * g.sf_sz);
  for (int slot = 0; slot < 2; slot++) {
    for (int l = 0; l < g.nsym; l++) {
      const float* s = in + 2 * (size_t)(slot * g.nsym + l) * g.nof_re;
      memset(X, 0, sizeof(cd) * N);
      for (int k = 0; k < g.nof_re; k++) {
        int f = k < g.nof_re / 2 ? N - g.nof_re / 2 + k : g.dc + k - g.nof_re / 2; /* ofdm.c:515-516 */
        X[f]  = s[2 * k] + I * s[2 * k + 1];
      }
      dft_f64(X, x, N, 1);
      const int special = g.mbsfn && slot == 0;
      int cp  = special ? g.mcp[l] : (l == 0 ? g.cp0 : g.cp1);
      int pos = special ? g.mpos[l] : slot * g.slot_sz + g.cp0 + l * (N + g.cp1);
      for (int n = 0; n < N; n++) {
        cd v = x[n];
        if (g.norm != 0.0) {
          v *= g.norm;
        }
        for (int rep = 0; rep < 2; rep++) { /* useful part, then its copy in the cyclic prefix (ofdm.c:532) */
          int q = rep == 0 ? pos + n : pos + n - N;
          if (rep == 1 && n < N - cp) {
            continue;
          }
          cd w = v;
          if (sh) {
            w = (cd)((float _Complex)w * sh[q]); /* ofdm.c:573-575, float product */
          }
          out[2 * q]     = (float)creal(w);
          out[2 * q + 1] = (float)cimag(w);
        }
      }
    }
  }
  free(X);
  free(x);
  free(sh);
  return 0;
}

/* ---- float32 iterative FFT used only for the CPU-baseline timing leg of bench.py (a plain, decent
 * scalar port: radix-2 Stockham with a radix-3 first stage when 3 | n).  Not used for parity. */
void orc_fft_f32(const float* in, float* out, int n, int backward)
{
  float _Complex* a = malloc(sizeof(float _Complex) * n);
  float _Complex* b = malloc(sizeof(float _Complex) * n);
  for (int i = 0; i < n; i++) {
    a[i] = in[2 * i] + I * in[2 * i + 1];
  }
  float sg = backward ? 1.0f : -1.0f;
  int   ns = 1;
  int   rem = n;
  while (rem > 1) {
    int r = (rem % 2 == 0) ? 2 : ((rem % 3 == 0) ? 3 : ((rem % 5 == 0) ? 5 : rem));
    int nb = n / r;
    for (int j = 0; j < nb; j++) {
      int            k  = j % ns;
      float _Complex v[8];
      if (r > 5) { /* generic prime: not needed for LTE/NR sizes */
        free(a);
        free(b);
        return;
      }
      for (int q = 0; q < r; q++) {
        float ang = sg * 2.0f * (float)M_PI * (float)(q * k) / (float)(ns * r);
        v[q]      = a[j + q * nb] * (cosf(ang) + I * sinf(ang));
      }
      int j0 = (j / ns) * ns * r + k;
      for (int q = 0; q < r; q++) {
        float _Complex acc = 0;
        for (int t = 0; t < r; t++) {
          float ang = sg * 2.0f * (float)M_PI * (float)((q * t) % r) / (float)r;
          acc += v[t] * (cosf(ang) + I * sinf(ang));
        }
        b[j0 + q * ns] = acc;
      }
    }
    float _Complex* t = a;
    a = b;
    b = t;
    ns *= r;
    rem /= r;
  }
  for (int i = 0; i < n; i++) {
    out[2 * i]     = crealf(a[i]);
    out[2 * i + 1] = cimagf(a[i]);
  }
  free(a);
  free(b);
}
