// ofdm_host.cpp -- host side of the OFDM (de)modulator: geometry, tables, batch object, srsran_ofdm_* ABI.
//
// Mirrors (interface + behaviour) lib/src/phy/dft/ofdm.c and the sizing helpers of
// lib/src/phy/common/phy_common.c:322-385 / lib/include/srsran/phy/common/phy_common.h:110-140.
#include "hip_common.h"
#include "ofdm_device.h"

#include <cmath>
#include <complex>
#include <vector>

using namespace phyhip;

// ------------------------------------------------------------------------------------------------ sizing helpers

static bool g_use_standard_rates =
#ifdef FORCE_STANDARD_RATE
    true;
#else
    false;
#endif

extern "C" void srsran_use_standard_symbol_size(bool enabled)
{
  g_use_standard_rates = enabled;
}

extern "C" int srsran_symbol_sz_power2(uint32_t nof_prb)
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

extern "C" int srsran_symbol_sz(uint32_t nof_prb)
{
  if (nof_prb == 0) {
    return SRSRAN_ERROR;
  }
  if (g_use_standard_rates) {
    return srsran_symbol_sz_power2(nof_prb);
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
  return SRSRAN_ERROR;
}

// The "standard symbol size" switch is PROCESS state of the reference's phy_common.c (srsran_use_standard_symbol_size, phy_common.c:31-35,
// 322-325), and an application that links this library keeps that file: it has a hundred other functions.  The library is linked
// -Bsymbolic, so a plain call from in here would bind to the copy above and its own flag -- and an application that switched its own
// copy to standard rates (pmch_file_test.c, sync_sl_test.c: 25 PRB -> 512 in the program, 384 in here) would get an OFDM object of another
// size than it asked for.  The default symbol size is therefore asked of the first definition in the process's global scope (the
// application's, when it has one; this library's otherwise -- e.g. when it was loaded privately by ctypes).
#include <dlfcn.h>
static int process_symbol_sz(uint32_t nof_prb)
{
  using fn_t        = int (*)(uint32_t);
  static const fn_t f = []() -> fn_t {
    // (through the main program's handle: RTLD_DEFAULT looked up from inside a -Bsymbolic library finds the library's own definition first)
    void* self = dlopen(nullptr, RTLD_LAZY);
    void* p    = self ? dlsym(self, "srsran_symbol_sz") : nullptr;
    return p ? reinterpret_cast<fn_t>(p) : &srsran_symbol_sz;
  }();
  return f(nof_prb);
}

// SRSRAN_CP_LEN (phy_common.h:125): float arithmetic on purpose
static int cp_len(uint32_t symbol_sz, int c)
{
  return (int)ceilf((((float)(c) * (symbol_sz)) / 2048.0f));
}

namespace {

struct Geometry {
  int   N = 0, nsym_slot = 0, cp0 = 0, cp1 = 0, nof_re = 0, slot_sz = 0, sf_sz = 0;
  int   dc = 0, win_n = 0;
  bool  norm = false;
  bool  shift_on = false;
  float freq_shift = 0.f;
  bool  mbsfn = false; // MBSFN subframe: slot 0 laid out by mbsfn_layout()
};

// symbol positions and CP lengths of the MBSFN slot; rx: ofdm.c:424-437, tx: ofdm.c:538-555.
// Returns false when the layout does not fit the subframe (region 0 on rx, region > 6)
bool mbsfn_layout(int N, int sf_sz, int region, bool tx, int (&pos)[6], int (&cp)[6], int* gap_start, int* gap_len)
{
  const int ext = cp_len(N, 512), n0 = cp_len(N, 160), n1 = cp_len(N, 144);
  // SRSRAN_NON_MBSFN_REGION_GUARD_LENGTH, phy_common.h:166
  const int guard = region == 1 ? ext - n0 : 2 * ext - n0 - n1;
  int       at    = 0;
  *gap_start = *gap_len = 0;
  for (int i = 0; i < 6; i++) {
    if (!tx) {
      if (i == region) {
        at += guard;
      }
      cp[i] = i >= region ? ext : (i == 0 ? n0 : n1);
      at += cp[i];
      pos[i] = at;
      at += N;
    } else {
      cp[i]  = i > region - 1 ? ext : (i == 0 ? n0 : n1);
      pos[i] = at + cp[i];
      at += N + cp[i];
      if (i == region - 1) {
        *gap_start = at;
        *gap_len   = guard;
        at += guard;
      }
    }
    if (pos[i] - cp[i] < 0 || pos[i] + N > sf_sz) {
      return false;
    }
  }
  return true;
}

void make_twiddles(int N, std::vector<std::complex<float>>& tw)
{
  tw.resize(N);
  for (int i = 0; i < N; i++) {
    double a = -2.0 * M_PI * (double)i / (double)N;
    tw[i]    = std::complex<float>((float)cos(a), (float)sin(a));
  }
}

// ofdm.c:130-138
void make_ramp(int N, uint32_t win_n, cf_t* ramp)
{
  for (int i = 0; i < N; i++) {
    // cexpf(I * M_PI * 2.0f * (float)n * (float)i / (float)N): phase evaluated in double, rounded to float
    double ph = M_PI * (double)2.0f * (double)(float)win_n * (double)(float)i / (double)(float)N;
    float  pf = (float)ph;
    ramp[i]   = cf_t(cosf(pf), sinf(pf));
  }
}

// ofdm.c:344-356
void make_shift(const Geometry& g, cf_t* tab)
{
  cf_t* ptr = tab;
  for (int n = 0; n < 2; n++) {
    for (int i = 0; i < g.nsym_slot; i++) {
      int cplen = i == 0 ? g.cp0 : g.cp1;
      for (int t = 0; t < g.N + cplen; t++) {
        double ph = 2.0 * M_PI * (double)((float)t - (float)cplen) * (double)g.freq_shift / (double)g.N;
        float  pf = (float)ph;
        ptr[t]    = cf_t(cosf(pf), sinf(pf));
      }
      ptr += g.N + cplen;
    }
  }
}

} // namespace

// ------------------------------------------------------------------------------------------------ batch object

struct srsran_hip_ofdm_batch {
  DeviceTag tag;
  Geometry g;
  int      region  = 2; // non-MBSFN region length of an MBSFN subframe (default of ofdm.c:198)
  bool     tx      = false;
  float2*  d_tw    = nullptr;
  float2*  d_shift = nullptr;
  float2*  d_ramp  = nullptr;
};

static int batch_build(srsran_hip_ofdm_batch_t** hh, const Geometry& g, bool tx, const cf_t* shift_tab, const cf_t* ramp_tab)
{
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  if (!ofdm::size_supported(g.N)) {
    set_error("OFDM symbol size %d is not supported by the HIP engine", g.N);
    return SRSRAN_ERROR;
  }
  if (g.nof_re > g.N - g.dc || g.nof_re <= 0 || (g.nof_re & 1)) {
    set_error("OFDM: nof_re=%d does not fit symbol size %d", g.nof_re, g.N);
    return SRSRAN_ERROR;
  }
  if (g.mbsfn && g.nsym_slot != 6) {
    set_error("OFDM: MBSFN subframes need the extended cyclic prefix");
    return SRSRAN_ERROR;
  }
  if (!tx && g.win_n > g.cp1) {
    set_error("OFDM: rx window offset of %d samples exceeds the cyclic prefix (%d)", g.win_n, g.cp1);
    return SRSRAN_ERROR;
  }
  auto* h = new srsran_hip_ofdm_batch;
  h->g    = g;
  h->tx   = tx;
  std::vector<std::complex<float>> tw;
  make_twiddles(g.N, tw);
  PHY_HIP_CHECK(hipMalloc(&h->d_tw, g.N * sizeof(float2)), SRSRAN_ERROR);
  PHY_HIP_CHECK(upload(h->d_tw, tw.data(), g.N * sizeof(float2)), SRSRAN_ERROR);
  if (g.shift_on) {
    PHY_HIP_CHECK(hipMalloc(&h->d_shift, g.sf_sz * sizeof(float2)), SRSRAN_ERROR);
    PHY_HIP_CHECK(upload(h->d_shift, shift_tab, g.sf_sz * sizeof(float2)), SRSRAN_ERROR);
  }
  if (!tx && g.win_n) {
    PHY_HIP_CHECK(hipMalloc(&h->d_ramp, g.N * sizeof(float2)), SRSRAN_ERROR);
    PHY_HIP_CHECK(upload(h->d_ramp, ramp_tab, g.N * sizeof(float2)), SRSRAN_ERROR);
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

// geometry of a freshly initialised object (ofdm.c:38-212 with max_prb == 0)
static int geometry_from_cfg(const srsran_ofdm_cfg_t* cfg, Geometry* g, uint32_t* win_n_out)
{
  uint32_t symbol_sz = cfg->symbol_sz;
  if (symbol_sz == 0) {
    int s = process_symbol_sz(cfg->nof_prb);
    if (s <= SRSRAN_SUCCESS) {
      fprintf(stderr, "Invalid number of PRB %d\n", cfg->nof_prb);
      return SRSRAN_ERROR;
    }
    symbol_sz = (uint32_t)s;
  }
  const bool normcp = cfg->cp == SRSRAN_CP_NORM;
  g->N         = (int)symbol_sz;
  g->nsym_slot = normcp ? 7 : 6;
  g->cp0       = normcp ? cp_len(symbol_sz, 160) : cp_len(symbol_sz, 512);
  g->cp1       = normcp ? cp_len(symbol_sz, 144) : cp_len(symbol_sz, 512);
  g->nof_re    = (int)cfg->nof_prb * 12;
  g->slot_sz   = (int)(symbol_sz * 15 / 2);
  g->sf_sz     = (int)(symbol_sz * 15);
  g->norm      = cfg->normalize;
  g->freq_shift = cfg->freq_shift_f;
  g->shift_on  = std::isnormal(cfg->freq_shift_f);
  g->dc        = ((!cfg->keep_dc) && !g->shift_on) ? 1 : 0;
  uint32_t win_n = 0;
  if (std::isnormal(cfg->rx_window_offset)) {
    float off = cfg->rx_window_offset;
    off       = off < 0 ? 0 : off;
    off       = off > 100 ? 100 : off;
    win_n     = (uint32_t)roundf((float)g->cp1 * off);
  }
  g->win_n = (int)win_n;
  g->mbsfn = cfg->sf_type == SRSRAN_SF_MBSFN;
  if (win_n_out) {
    *win_n_out = win_n;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_ofdm_batch_create(srsran_hip_ofdm_batch_t** hh, const srsran_ofdm_cfg_t* cfg, srsran_dft_dir_t dir)
{
  if (!hh || !cfg) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  Geometry g;
  if (geometry_from_cfg(cfg, &g, nullptr)) {
    return SRSRAN_ERROR;
  }
  const bool tx = dir == SRSRAN_DFT_BACKWARD;
  if (tx) {
    g.win_n = 0;
  }
  std::vector<cf_t> shift(g.shift_on ? g.sf_sz : 0), ramp(g.win_n ? g.N : 0);
  if (g.shift_on) {
    make_shift(g, shift.data());
  }
  if (g.win_n) {
    make_ramp(g.N, (uint32_t)g.win_n, ramp.data());
  }
  return batch_build(hh, g, tx, shift.data(), ramp.data());
}

extern "C" void srsran_hip_ofdm_batch_free(srsran_hip_ofdm_batch_t* h)
{
  if (!h) {
    return;
  }
  (void)hipFree(h->d_tw);
  (void)hipFree(h->d_shift);
  (void)hipFree(h->d_ramp);
  delete h;
}

extern "C" uint32_t srsran_hip_ofdm_batch_sf_sz(srsran_hip_ofdm_batch_t* h)
{
  return h ? (uint32_t)h->g.sf_sz : 0;
}

extern "C" uint32_t srsran_hip_ofdm_batch_sf_re(srsran_hip_ofdm_batch_t* h)
{
  return h ? (uint32_t)(h->g.nof_re * 2 * h->g.nsym_slot) : 0;
}

static int batch_run(srsran_hip_ofdm_batch_t* h, const void* d_in, void* d_out, uint32_t n_sf, bool tx, bool with_shift,
                     hipStream_t stream, bool with_ramp = true)
{
  TraceRange trace_("srsran_hip_ofdm_batch");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_ofdm_batch", SRSRAN_ERROR);
  }
  if (h && n_sf == 0 && h->tx == tx) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!h || !d_in || !d_out || h->tx != tx) {
    set_error("ofdm batch: invalid arguments or wrong direction");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  const Geometry& g = h->g;
  ofdm::Params    p;
  p.in          = d_in;
  p.out         = d_out;
  p.twiddle     = h->d_tw;
  p.shift       = with_shift ? h->d_shift : nullptr;
  p.ramp        = (tx || !with_ramp) ? nullptr : h->d_ramp;
  p.n_sym_total = (long)n_sf * 2 * g.nsym_slot;
  p.N           = g.N;
  p.nsym_sf     = 2 * g.nsym_slot;
  p.slot_sz     = g.slot_sz;
  p.sf_sz       = g.sf_sz;
  p.cp0         = g.cp0;
  p.cp1         = g.cp1;
  p.nof_re      = g.nof_re;
  p.dc          = g.dc;
  p.win_n       = tx ? 0 : g.win_n;
  p.spw         = 1;
  p.norm        = g.norm ? 1.0f / sqrtf((float)g.N) : 0.0f;
  p.mbsfn       = g.mbsfn ? 1 : 0;
  if (g.mbsfn) {
    int gs, gl;
    if (!mbsfn_layout(g.N, g.sf_sz, h->region, tx, p.mpos, p.mcp, &gs, &gl)) {
      set_error("ofdm batch: non-MBSFN region of %d symbols does not fit the subframe", h->region);
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
  }
  PHY_HIP_CHECK(ofdm::launch(p, tx, stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_ofdm_batch_set_non_mbsfn_region(srsran_hip_ofdm_batch_t* h, uint8_t non_mbsfn_region)
{
  if (!h || !h->g.mbsfn) {
    set_error("ofdm batch: not an MBSFN object");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  h->region = non_mbsfn_region;
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_ofdm_batch_rx(srsran_hip_ofdm_batch_t* h, const cf_t* d_in, cf_t* d_out, uint32_t n_sf, void* stream)
{
  return batch_run(h, d_in, d_out, n_sf, false, true, (hipStream_t)stream);
}

extern "C" int srsran_hip_ofdm_batch_tx(srsran_hip_ofdm_batch_t* h, const cf_t* d_in, cf_t* d_out, uint32_t n_sf, void* stream)
{
  return batch_run(h, d_in, d_out, n_sf, true, true, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ handle ABI

namespace {
struct OfdmCtx {
  DeviceTag                tag;
  srsran_hip_ofdm_batch_t* b      = nullptr;
  hipStream_t              stream = nullptr;
  float2*                  d_time = nullptr; // sf_sz
  float2*                  d_re   = nullptr; // nof_re * nsym
  cf_t*                    h_time = nullptr; // pinned
  cf_t*                    h_re   = nullptr; // pinned
  size_t                   cap_time = 0, cap_re = 0;
  srsran_dft_dir_t         dir = SRSRAN_DFT_FORWARD;
};

OfdmCtx* ctx_raw(srsran_ofdm_t* q)
{
  return reinterpret_cast<OfdmCtx*>(q->tmp);
}
// nullptr (error reported) when the object lives on another device than the calling thread's
OfdmCtx* ctx_of(srsran_ofdm_t* q)
{
  OfdmCtx* c = ctx_raw(q);
  return (c && !check_device(c->tag, "srsran_ofdm")) ? nullptr : c;
}

void ctx_free(OfdmCtx* c)
{
  if (!c) {
    return;
  }
  srsran_hip_ofdm_batch_free(c->b);
  (void)hipFree(c->d_time);
  (void)hipFree(c->d_re);
  (void)hipHostFree(c->h_time);
  (void)hipHostFree(c->h_re);
  if (c->stream) {
    (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

// the state machine of ofdm_init_mbsfn_ (ofdm.c:38-212), including its reconfiguration quirks
int ofdm_init_(srsran_ofdm_t* q, srsran_ofdm_cfg_t* cfg, srsran_dft_dir_t dir)
{
  if (cfg->symbol_sz == 0) {
    int s = process_symbol_sz(cfg->nof_prb);
    if (s <= SRSRAN_SUCCESS) {
      fprintf(stderr, "Invalid number of PRB %d\n", cfg->nof_prb);
      return SRSRAN_ERROR;
    }
    cfg->symbol_sz = (uint32_t)s;
  }
  if (q->max_prb > 0) {
    // already initialised: only the resizing parameters change (:52-56)
    q->cfg.cp        = cfg->cp;
    q->cfg.nof_prb   = cfg->nof_prb;
    q->cfg.symbol_sz = cfg->symbol_sz;
  } else {
    q->cfg = *cfg;
  }
  const uint32_t    symbol_sz = q->cfg.symbol_sz;
  const srsran_cp_t cp        = q->cfg.cp;
  const bool        normcp    = cp == SRSRAN_CP_NORM;

  q->nof_symbols       = normcp ? 7 : 6;
  q->nof_symbols_mbsfn = 6;
  q->nof_re            = cfg->nof_prb * 12;
  q->nof_guards        = (q->cfg.symbol_sz - q->nof_re) / 2U;
  q->slot_sz           = symbol_sz * 15 / 2;
  q->sf_sz             = symbol_sz * 15;

  // the single-symbol plan of the reference only carries flags here (:72-85,186,207-209)
  if (!q->fft_plan.size) {
    q->fft_plan.init_size = (int)symbol_sz;
    q->fft_plan.dir       = dir;
    q->fft_plan.forward   = dir == SRSRAN_DFT_FORWARD;
    q->fft_plan.mode      = SRSRAN_DFT_COMPLEX;
  } else if ((int)symbol_sz > q->fft_plan.init_size) {
    fprintf(stderr, "DFT: Error calling replan: new_dft_points (%d) must be lower or equal dft_size passed initially (%d)\n",
            symbol_sz, q->fft_plan.init_size);
    return SRSRAN_ERROR;
  }
  q->fft_plan.size = (int)symbol_sz;

  OfdmCtx* c = ctx_of(q);
  if (!c) {
    if (!device_available()) {
      return SRSRAN_ERROR;
    }
    c      = new OfdmCtx;
    c->dir = dir;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return SRSRAN_ERROR;
    }
    q->tmp = reinterpret_cast<cf_t*>(c);
  }
  if (q->cfg.nof_prb > q->max_prb) {
    // (re)allocate the host tables (:88-116); the reference leaks window_offset_buffer here, we do not
    free(q->shift_buffer);
    free(q->window_offset_buffer);
    q->shift_buffer         = (cf_t*)calloc(q->sf_sz, sizeof(cf_t));
    q->window_offset_buffer = (cf_t*)calloc(q->sf_sz, sizeof(cf_t));
    if (!q->shift_buffer || !q->window_offset_buffer) {
      perror("malloc");
      return SRSRAN_ERROR;
    }
    q->max_prb = cfg->nof_prb;
  }

  const uint32_t nof_prb = q->cfg.nof_prb;
  const int      cp1     = normcp ? cp_len(symbol_sz, 160) : cp_len(symbol_sz, 512);
  const int      cp2     = normcp ? cp_len(symbol_sz, 144) : cp_len(symbol_sz, 512);

  // window offset is taken from the cfg of THIS call (:126-139): a set_prb() call keeps the old
  // window_offset_n and does not rebuild the ramp
  if (std::isnormal(cfg->rx_window_offset)) {
    cfg->rx_window_offset = cfg->rx_window_offset < 0 ? 0 : cfg->rx_window_offset;
    cfg->rx_window_offset = cfg->rx_window_offset > 100 ? 100 : cfg->rx_window_offset;
    q->window_offset_n    = (uint32_t)roundf((float)cp2 * cfg->rx_window_offset);
    make_ramp((int)symbol_sz, q->window_offset_n, q->window_offset_buffer);
  }

  // the reference zeroes the caller's input buffer at (re)initialisation (:142-147)
  if (q->cfg.in_buffer) {
    size_t n = dir == SRSRAN_DFT_BACKWARD ? (size_t)2 * nof_prb * 12 * q->nof_symbols : (size_t)q->sf_sz;
    memset((void*)q->cfg.in_buffer, 0, n * sizeof(cf_t));
  }

  q->fft_plan.mirror = true;

  if (q->cfg.sf_type == SRSRAN_SF_MBSFN) { // :196-201
    q->mbsfn_subframe   = true;
    q->non_mbsfn_region = 2;
  } else {
    q->mbsfn_subframe = false;
  }

  // :205-209
  if (srsran_ofdm_set_freq_shift(q, q->cfg.freq_shift_f)) {
    return SRSRAN_ERROR;
  }
  q->fft_plan.norm = q->cfg.normalize;
  q->fft_plan.dc   = (!cfg->keep_dc) && (!std::isnormal(q->cfg.freq_shift_f));
  (void)cp1;
  return SRSRAN_SUCCESS;
}

// (re)build the device object from the CURRENT handle state; called lazily by the run functions so that
// srsran_ofdm_set_freq_shift / set_normalize after init are honoured like in the reference
int ctx_sync(srsran_ofdm_t* q)
{
  OfdmCtx* c = ctx_of(q);
  if (!c) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ofdm: object not initialised\n");
    return SRSRAN_ERROR;
  }
  bind_thread();
  const bool normcp = q->cfg.cp == SRSRAN_CP_NORM;
  Geometry   g;
  g.N          = (int)q->cfg.symbol_sz;
  g.nsym_slot  = (int)q->nof_symbols;
  g.cp0        = normcp ? cp_len(g.N, 160) : cp_len(g.N, 512);
  g.cp1        = normcp ? cp_len(g.N, 144) : cp_len(g.N, 512);
  g.nof_re     = (int)q->nof_re;
  g.slot_sz    = (int)q->slot_sz;
  g.sf_sz      = (int)q->sf_sz;
  g.dc         = q->fft_plan.dc ? 1 : 0;
  g.win_n      = c->dir == SRSRAN_DFT_FORWARD ? (int)q->window_offset_n : 0;
  g.norm       = q->fft_plan.norm;
  g.freq_shift = q->cfg.freq_shift_f;
  g.shift_on   = std::isnormal(q->cfg.freq_shift_f);
  g.mbsfn      = q->mbsfn_subframe;
  bool same = c->b && c->b->g.mbsfn == g.mbsfn && c->b->g.N == g.N && c->b->g.nsym_slot == g.nsym_slot && c->b->g.nof_re == g.nof_re &&
              c->b->g.dc == g.dc && c->b->g.win_n == g.win_n && c->b->g.norm == g.norm &&
              c->b->g.shift_on == g.shift_on && c->b->g.freq_shift == g.freq_shift;
  if (same) {
    c->b->region = q->non_mbsfn_region;
    return SRSRAN_SUCCESS;
  }
  srsran_hip_ofdm_batch_free(c->b);
  c->b = nullptr;
  if (batch_build(&c->b, g, c->dir == SRSRAN_DFT_BACKWARD, q->shift_buffer, q->window_offset_buffer)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ofdm: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  c->b->region = q->non_mbsfn_region;
  const size_t nt = (size_t)g.sf_sz, nr = (size_t)g.nof_re * 2 * g.nsym_slot;
  if (nt > c->cap_time) {
    (void)hipFree(c->d_time);
    (void)hipHostFree(c->h_time);
    PHY_HIP_CHECK(hipMalloc(&c->d_time, nt * sizeof(float2)), SRSRAN_ERROR);
    PHY_HIP_CHECK(host_image_alloc(&c->h_time, nt * sizeof(cf_t)), SRSRAN_ERROR);
    c->cap_time = nt;
  }
  if (nr > c->cap_re) {
    (void)hipFree(c->d_re);
    (void)hipHostFree(c->h_re);
    PHY_HIP_CHECK(hipMalloc(&c->d_re, nr * sizeof(float2)), SRSRAN_ERROR);
    PHY_HIP_CHECK(host_image_alloc(&c->h_re, nr * sizeof(cf_t)), SRSRAN_ERROR);
    c->cap_re = nr;
  }
  return SRSRAN_SUCCESS;
}

void ofdm_free_(srsran_ofdm_t* q)
{
  ctx_free(ctx_raw(q));
  free(q->shift_buffer);
  free(q->window_offset_buffer);
  memset(q, 0, sizeof(srsran_ofdm_t)); // ofdm.c:240
}

void rx_run(srsran_ofdm_t* q, cf_t* input, cf_t* output, bool with_ramp)
{
  if (ctx_sync(q)) {
    return;
  }
  OfdmCtx*       c  = ctx_of(q);
  const size_t   nt = q->sf_sz, nr = (size_t)q->nof_re * 2 * q->nof_symbols;
  const bool     sh = std::isnormal(q->cfg.freq_shift_f);
  memcpy(c->h_time, input, nt * sizeof(cf_t));
  // One subframe per call: the demodulator reads every sample once and writes every resource element once, so it works on the PINNED HOST
  // images themselves (mapped into the device's address space) -- no copy operation on either side of the kernel: a copy operation costs
  // 6-9 us whatever its size, the 380 KB of a 20 MHz subframe cross the bus in about as much (tools/probe/roundtrip_probe.hip).
  float2* t_in = reinterpret_cast<float2*>(c->h_time);
  if (sh) {
    // ofdm.c:455-457 multiplies the caller's input buffer in place; reproduce the side effect (product kept on the device for the transform)
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->d_time, c->h_time, nt * sizeof(cf_t), hipMemcpyHostToDevice, c->stream));
    PHY_HIP_CHECK_VOID(ofdm::launch_prod_ccc(c->d_time, c->b->d_shift, c->d_time, (int)nt, c->stream));
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->h_time, c->d_time, nt * sizeof(cf_t), hipMemcpyDeviceToHost, c->stream));
    t_in = c->d_time;
  }
  if (batch_run(c->b, t_in, reinterpret_cast<float2*>(c->h_re), 1, false, false, c->stream, with_ramp)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ofdm_rx_sf: %s\n", get_error());
    return;
  }
  PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
  if (sh) {
    memcpy(input, c->h_time, nt * sizeof(cf_t));
  }
  memcpy(output, c->h_re, nr * sizeof(cf_t));
}

} // namespace

extern "C" int srsran_ofdm_rx_init(srsran_ofdm_t* q, srsran_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb)
{
  memset(q, 0, sizeof(srsran_ofdm_t));
  srsran_ofdm_cfg_t cfg = {};
  cfg.cp                = cp;
  cfg.in_buffer         = in_buffer;
  cfg.out_buffer        = out_buffer;
  cfg.nof_prb           = max_prb;
  cfg.sf_type           = SRSRAN_SF_NORM;
  return ofdm_init_(q, &cfg, SRSRAN_DFT_FORWARD);
}

extern "C" int srsran_ofdm_rx_init_mbsfn(srsran_ofdm_t* q, srsran_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb)
{
  memset(q, 0, sizeof(srsran_ofdm_t));
  srsran_ofdm_cfg_t cfg = {};
  cfg.cp                = cp;
  cfg.in_buffer         = in_buffer;
  cfg.out_buffer        = out_buffer;
  cfg.nof_prb           = max_prb;
  cfg.sf_type           = SRSRAN_SF_MBSFN;
  return ofdm_init_(q, &cfg, SRSRAN_DFT_FORWARD);
}

extern "C" int srsran_ofdm_tx_init(srsran_ofdm_t* q, srsran_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t max_prb)
{
  memset(q, 0, sizeof(srsran_ofdm_t));
  srsran_ofdm_cfg_t cfg = {};
  cfg.cp                = cp;
  cfg.in_buffer         = in_buffer;
  cfg.out_buffer        = out_buffer;
  cfg.nof_prb           = max_prb;
  cfg.sf_type           = SRSRAN_SF_NORM;
  return ofdm_init_(q, &cfg, SRSRAN_DFT_BACKWARD);
}

extern "C" int srsran_ofdm_tx_init_mbsfn(srsran_ofdm_t* q, srsran_cp_t cp, cf_t* in_buffer, cf_t* out_buffer, uint32_t nof_prb)
{
  memset(q, 0, sizeof(srsran_ofdm_t));
  srsran_ofdm_cfg_t cfg = {};
  cfg.cp                = cp;
  cfg.in_buffer         = in_buffer;
  cfg.out_buffer        = out_buffer;
  cfg.nof_prb           = nof_prb;
  cfg.sf_type           = SRSRAN_SF_MBSFN;
  return ofdm_init_(q, &cfg, SRSRAN_DFT_BACKWARD);
}

extern "C" int srsran_ofdm_tx_init_cfg(srsran_ofdm_t* q, srsran_ofdm_cfg_t* cfg)
{
  return ofdm_init_(q, cfg, SRSRAN_DFT_BACKWARD);
}

extern "C" int srsran_ofdm_rx_init_cfg(srsran_ofdm_t* q, srsran_ofdm_cfg_t* cfg)
{
  return ofdm_init_(q, cfg, SRSRAN_DFT_FORWARD);
}

extern "C" int srsran_ofdm_rx_set_prb(srsran_ofdm_t* q, srsran_cp_t cp, uint32_t nof_prb)
{
  srsran_ofdm_cfg_t cfg = {};
  cfg.cp                = cp;
  cfg.nof_prb           = nof_prb;
  return ofdm_init_(q, &cfg, SRSRAN_DFT_FORWARD);
}

extern "C" int srsran_ofdm_tx_set_prb(srsran_ofdm_t* q, srsran_cp_t cp, uint32_t nof_prb)
{
  srsran_ofdm_cfg_t cfg = {};
  cfg.cp                = cp;
  cfg.nof_prb           = nof_prb;
  return ofdm_init_(q, &cfg, SRSRAN_DFT_BACKWARD);
}

extern "C" void srsran_ofdm_rx_free(srsran_ofdm_t* q)
{
  ofdm_free_(q);
}

extern "C" void srsran_ofdm_tx_free(srsran_ofdm_t* q)
{
  ofdm_free_(q);
}

extern "C" int srsran_ofdm_set_freq_shift(srsran_ofdm_t* q, float freq_shift)
{
  q->cfg.freq_shift_f = freq_shift;
  if (!std::isnormal(q->cfg.freq_shift_f)) {
    q->fft_plan.dc = true; // ofdm.c:339-342
    return SRSRAN_SUCCESS;
  }
  if (!q->shift_buffer) {
    return SRSRAN_ERROR;
  }
  Geometry   g;
  const bool normcp = q->cfg.cp == SRSRAN_CP_NORM;
  g.N          = (int)q->cfg.symbol_sz;
  g.nsym_slot  = (int)q->nof_symbols;
  g.cp0        = normcp ? cp_len(g.N, 160) : cp_len(g.N, 512);
  g.cp1        = normcp ? cp_len(g.N, 144) : cp_len(g.N, 512);
  g.freq_shift = freq_shift;
  make_shift(g, q->shift_buffer);
  OfdmCtx* c = ctx_of(q);
  if (c && c->b) {
    // force ctx_sync to rebuild the device tables
    srsran_hip_ofdm_batch_free(c->b);
    c->b = nullptr;
  }
  q->fft_plan.dc = false; // ofdm.c:359
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_ofdm_set_normalize(srsran_ofdm_t* q, bool normalize_enable)
{
  q->fft_plan.norm = normalize_enable;
}

extern "C" void srsran_ofdm_set_non_mbsfn_region(srsran_ofdm_t* q, uint8_t non_mbsfn_region)
{
  q->non_mbsfn_region = non_mbsfn_region;
}

extern "C" void srsran_ofdm_rx_sf(srsran_ofdm_t* q)
{
  rx_run(q, q->cfg.in_buffer, q->cfg.out_buffer, true);
}

extern "C" void srsran_ofdm_rx_sf_ng(srsran_ofdm_t* q, cf_t* input, cf_t* output)
{
  // The reference's non-guru path (ofdm.c:468-482, 368-383) runs the single-symbol plan (mirror=true,
  // dc, norm) and copies &tmp[nof_guards]: same RE order and scaling as the guru path.  It moves the
  // FFT window by window_offset_n but does NOT apply the compensating phase ramp (:375-377 vs :405-407);
  // that quirk is kept.
  if (q->mbsfn_subframe) {
    // ofdm.c:477-480: the MBSFN branch works on the buffers given at init, whatever the arguments are
    rx_run(q, q->cfg.in_buffer, q->cfg.out_buffer, true);
    return;
  }
  rx_run(q, input, output, false);
}

extern "C" void srsran_ofdm_tx_sf(srsran_ofdm_t* q)
{
  if (ctx_sync(q)) {
    return;
  }
  OfdmCtx*     c  = ctx_of(q);
  const size_t nt = q->sf_sz, nr = (size_t)q->nof_re * 2 * q->nof_symbols;
  memcpy(c->h_re, q->cfg.in_buffer, nr * sizeof(cf_t));
  // (as rx_run: the modulator reads and writes the pinned host images directly)
  if (batch_run(c->b, reinterpret_cast<float2*>(c->h_re), reinterpret_cast<float2*>(c->h_time), 1, true, true, c->stream)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ofdm_tx_sf: %s\n", get_error());
    return;
  }
  PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
  if (q->mbsfn_subframe) {
    // the samples between the non-MBSFN and the MBSFN region are not written (ofdm.c:551-553)
    int pos[6], cp[6], gs = 0, gl = 0;
    mbsfn_layout((int)q->cfg.symbol_sz, (int)q->sf_sz, q->non_mbsfn_region, true, pos, cp, &gs, &gl);
    memcpy(q->cfg.out_buffer, c->h_time, (size_t)gs * sizeof(cf_t));
    memcpy(q->cfg.out_buffer + gs + gl, c->h_time + gs + gl, (nt - gs - gl) * sizeof(cf_t));
  } else {
    memcpy(q->cfg.out_buffer, c->h_time, nt * sizeof(cf_t));
  }
}
