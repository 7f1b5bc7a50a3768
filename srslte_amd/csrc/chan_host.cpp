// chan_host.cpp -- one device call per grant (include/srsran_amd/phy_chan_abi.h): the stages between the resource grid and the transport
// block of srsran_pusch_decode (pusch.c:358-478), srsran_pdsch_decode / _encode (pdsch.c:662-760, 949-1015) and srsran_ulsch_encode
// (sch.c:1194) chained on the calling thread's transport-block stream, nothing but the inputs and the results crossing the bus.
#include "chan_device.h"
#include "hip_common.h"
#include "modem_device.h"
#include "sch_stage.h"
#include "srsran_amd/phy_chan_abi.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

using namespace phyhip;

namespace {

inline uint32_t qm_of(uint32_t mod)
{
  return mod == 0 ? 1u : 2u * mod;
}
inline uint32_t qm_rm(const srsran_hip_grant_tb_t& tb) // what decode_tb / encode_tb get as Qm (sch.c:590,632)
{
  return qm_of(tb.mod) * (tb.nl ? tb.nl : 1u);
}

// per calling thread: pinned images the kernels read the grant's symbols / channel estimates from and write transmit symbols into, device
// scratch between the front-end kernels, the transform plans of the allocation sizes seen so far
struct ChanStage {
  uint8_t* pin     = nullptr;
  size_t   pin_cap = 0;
  uint8_t* dev     = nullptr;
  size_t   dev_cap = 0;
  std::map<uint32_t, srsran_hip_dft_batch_t*> idft; // L_prb -> backward, normalised plan of 12 L_prb points (srsran_dft_precoding_init_rx)
  ~ChanStage()
  {
    for (auto& kv : idft) {
      srsran_hip_dft_batch_free(kv.second);
    }
    (void)hipFree(dev);
    (void)hipHostFree(pin);
  }
  bool grow(size_t need_pin, size_t need_dev)
  {
    if (need_pin > pin_cap) {
      (void)hipHostFree(pin);
      pin     = nullptr;
      pin_cap = 0;
      if (host_image_alloc(&pin, need_pin + need_pin / 2) != hipSuccess) {
        return false;
      }
      pin_cap = need_pin + need_pin / 2;
    }
    if (need_dev > dev_cap) {
      (void)hipFree(dev);
      dev     = nullptr;
      dev_cap = 0;
      if (hipMalloc((void**)&dev, need_dev + need_dev / 2) != hipSuccess) {
        return false;
      }
      dev_cap = need_dev + need_dev / 2;
    }
    return true;
  }
  srsran_hip_dft_batch_t* plan(uint32_t L_prb)
  {
    auto it = idft.find(L_prb);
    if (it != idft.end()) {
      return it->second;
    }
    srsran_hip_dft_batch_t* h = nullptr;
    if (srsran_hip_dft_batch_create(&h, (int)(12 * L_prb), SRSRAN_DFT_BACKWARD, false, false, true) != SRSRAN_SUCCESS) {
      return nullptr;
    }
    idft[L_prb] = h;
    return h;
  }
};

ChanStage& stage()
{
  static thread_local StageRef<ChanStage> r;
  return r.get();
}

inline size_t al256(size_t v)
{
  return (v + 255) & ~(size_t)255;
}

bool tb_valid(const srsran_hip_grant_tb_t& tb, const char* who)
{
  if (tb.mod > SRSRAN_MOD_256QAM || tb.nl > 2 || tb.nof_re == 0 || tb.tbs == 0 || (tb.tbs & 7u) || tb.rv > 3 ||
      (uint64_t)tb.nof_re * qm_of(tb.mod) > SRSRAN_HIP_SEQUENCE_MAX_LEN) {
    set_error("%s: invalid grant (mod %u, %u REs, tbs %u, rv %u)", who, tb.mod, tb.nof_re, tb.tbs, tb.rv);
    fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
    return false;
  }
  return true;
}

// the receive front end of one grant, enqueued on `st`: [equaliser] -> [transform de-precoding] -> demodulator + descrambler (+ UL channel
// de-interleaver in its store) -> d_e.  p_sym / p_ce: the grant's REs in the pinned image; d_x / d_z: device scratch of nof_re points each.
bool enqueue_rx_front(hipStream_t st, const srsran_hip_grant_tb_t& tb, const uint8_t* p_sym, const uint8_t* p_ce, float scaling, float noise, uint32_t L_prb,
                      uint32_t nof_symb, uint8_t* d_x, uint8_t* d_z, srsran_hip_dft_batch_t* plan, void* d_e)
{
  const uint8_t* cur = p_sym;
  if (p_ce) {
    if (modem::launch_eq(p_sym, p_ce, d_x, nullptr, tb.nof_re, scaling, noise, st) != hipSuccess) {
      set_error("grant front end: equaliser launch failed");
      return false;
    }
    cur = d_x;
  }
  if (L_prb) {
    if (srsran_hip_dft_batch_run(plan, (const cf_t*)cur, (cf_t*)d_z, nof_symb, st) != SRSRAN_SUCCESS) {
      return false;
    }
    cur = d_z;
  }
  modem::Params p;
  if (!modem::params_for(p, tb.llr_is_8bit ? modem::LLR_I8 : modem::LLR_I16)) {
    return false;
  }
  p.in      = cur;
  p.out     = d_e;
  p.single  = modem::Job{tb.mod, tb.nof_re, 0, 0, tb.seed, 1u, 0, modem::tiles_of(tb.mod, tb.nof_re), L_prb ? 12 * L_prb : 0u, L_prb ? nof_symb : 0u};
  p.n_jobs  = 1;
  p.n_tiles = p.single.ntiles;
  if (modem::launch(p, st) != hipSuccess) {
    set_error("grant front end: demodulator launch failed");
    return false;
  }
  return true;
}

// srsran_vec_avg_power_cf over n points (a measurement: plain left-to-right float sums, not the reference's SIMD order)
float avg_power(const float* x, size_t n)
{
  double acc = 0;
  for (size_t i = 0; i < 2 * n; i++) {
    acc += (double)x[i] * x[i];
  }
  return n ? (float)(acc / (double)n) : 0.f;
}

struct PuschPlan { // one grant of a (multi-)call
  uint32_t                nof_symb = 0;
  size_t                  o_sym = 0, o_ce = 0, o_x = 0, o_z = 0; // byte offsets in the pinned (symbols, estimates) and device (equalised, de-precoded) images
  srsran_cbsegm_t         seg;
  srsran_hip_sch_head_t   head;
  srsran_hip_dft_batch_t* plan = nullptr;
  sch::FrontEnd           front;
};

} // namespace

// ------------------------------------------------------------------------------------------------ PUSCH receive

extern "C" int srsran_hip_pusch_decode_multi(uint32_t n, const srsran_hip_pusch_rx_t* g, const cf_t* const* sf_symbols, const cf_t* const* ce,
                                             srsran_softbuffer_rx_t* const* softbuffers, uint8_t* const* data, srsran_hip_grant_res_t* res)
{
  TraceRange trace_("srsran_hip_pusch_decode");
  if (n == 0) {
    return SRSRAN_SUCCESS;
  }
  if (!g || !sf_symbols || !ce || !softbuffers || !data || !res) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_pusch_decode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  bind_thread();
  ChanStage&             s = stage();
  std::vector<PuschPlan> pl(n);
  size_t                 pin_need = 0, dev_need = 0;
  for (uint32_t i = 0; i < n; i++) {
    const srsran_hip_pusch_rx_t& x = g[i];
    res[i] = {0, 0.f, NAN};
    if (!tb_valid(x.tb, "srsran_hip_pusch_decode") || !sf_symbols[i] || !ce[i] || !softbuffers[i] || !data[i]) {
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    const uint32_t nsymb = 2 * (x.cp_nsymb - 1) - (x.shortened ? 1u : 0u);
    if ((x.cp_nsymb != 7 && x.cp_nsymb != 6) || x.L_prb == 0 || !srsran_dft_precoding_valid_prb(x.L_prb) || x.n_prb_tilde[0] + x.L_prb > x.cell_nof_prb ||
        x.n_prb_tilde[1] + x.L_prb > x.cell_nof_prb || x.tb.nof_re != nsymb * 12 * x.L_prb || x.tb.mod < SRSRAN_MOD_QPSK || x.tb.mod > SRSRAN_MOD_64QAM) {
      set_error("srsran_hip_pusch_decode: grant %u: allocation (%u PRB at %u / %u of %u, %u REs, mod %u) is not a PUSCH allocation", i, x.L_prb, x.n_prb_tilde[0],
                x.n_prb_tilde[1], x.cell_nof_prb, x.tb.nof_re, x.tb.mod);
      fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    PuschPlan& p = pl[i];
    p.nof_symb   = nsymb;
    if (srsran_cbsegm(&p.seg, x.tb.tbs) != SRSRAN_SUCCESS) {
      fprintf(stderr, "Error computing segmentation for TBS=%d\n", x.tb.tbs); // sch.c:1133-1136
      return SRSRAN_ERROR;
    }
    p.plan = s.plan(x.L_prb);
    if (!p.plan) {
      return SRSRAN_ERROR;
    }
  }
  // Layout: the grants' REs packed one behind the other, ordered by allocation size -- four arrays of the same shape (symbols and estimates in the pinned
  // image, equalised and de-precoded symbols on the device), so that ONE equaliser launch, one transform launch per allocation size and ONE demodulator
  // launch serve all grants of the call (a launch costs 4-5 us whatever it carries: 25 grants x 3 kernels were 330 of the call's 390 us).
  std::vector<uint32_t> ord(n);
  for (uint32_t i = 0; i < n; i++) {
    ord[i] = i;
  }
  std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return g[a].L_prb < g[b].L_prb; });
  size_t tot = 0, tiles = 0;
  for (uint32_t k = 0; k < n; k++) {
    PuschPlan& p = pl[ord[k]];
    p.o_sym = p.o_x = tot;
    tot += (size_t)g[ord[k]].tb.nof_re * sizeof(cf_t);
    tiles += modem::tiles_of(g[ord[k]].tb.mod, g[ord[k]].tb.nof_re);
  }
  const size_t region = al256(tot);
  for (uint32_t i = 0; i < n; i++) {
    pl[i].o_ce = region + pl[i].o_sym;
    pl[i].o_z  = region + pl[i].o_x;
  }
  const size_t o_eqj = 2 * region, o_mj = al256(o_eqj + n * sizeof(modem::EqJob)), o_tj = al256(o_mj + n * sizeof(modem::Job));
  pin_need = al256(o_tj + tiles * sizeof(uint32_t));
  dev_need = 2 * region;
  if (!s.grow(pin_need, dev_need)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_pusch_decode: staging allocation failed\n");
    return SRSRAN_ERROR;
  }
  std::vector<sch::TbItem> items(n);
  for (uint32_t i = 0; i < n; i++) {
    const srsran_hip_pusch_rx_t& x = g[i];
    PuschPlan&                   p = pl[i];
    // pusch.c:48-104 (pusch_get): the allocation's 12 L_prb sub-carriers of every symbol but the slot's reference symbol (and the SRS symbol)
    const uint32_t L_ref = x.cp_nsymb == 7 ? 3 : 2;
    const size_t   wid   = (size_t)x.L_prb * 12 * sizeof(cf_t);
    size_t         at    = 0;
    for (uint32_t slot = 0; slot < 2; slot++) {
      const uint32_t nl = x.cp_nsymb - ((x.shortened && slot == 1) ? 1u : 0u);
      for (uint32_t l = 0; l < nl; l++) {
        if (l == L_ref) {
          continue;
        }
        const size_t idx = ((size_t)(l + slot * x.cp_nsymb) * x.cell_nof_prb + x.n_prb_tilde[slot]) * 12;
        memcpy(s.pin + p.o_sym + at, sf_symbols[i] + idx, wid);
        memcpy(s.pin + p.o_ce + at, ce[i] + idx, wid);
        at += wid;
      }
    }
    if (x.meas_epre) {
      res[i].epre = avg_power(reinterpret_cast<const float*>(s.pin + p.o_sym), x.tb.nof_re); // pusch.c:397-401
    }
    p.head = {x.tb.max_nof_iterations, 0.f, x.tb.llr_is_8bit != 0};
    const srsran_hip_grant_tb_t tb = x.tb;
    const uint8_t *psym = s.pin + p.o_sym, *pce = s.pin + p.o_ce;
    uint8_t *      dx = s.dev + p.o_x, *dz = s.dev + p.o_z;
    const float    noise = x.noise_estimate;
    const uint32_t L_prb = x.L_prb, nsymb = p.nof_symb;
    auto*          plan  = p.plan;
    p.front = [=](hipStream_t st, void* d_e) { return enqueue_rx_front(st, tb, psym, pce, 1.0f, noise, L_prb, nsymb, dx, dz, plan, d_e); };
    items[i] = {&p.head, softbuffers[i], &p.seg, qm_rm(x.tb), x.tb.rv, x.tb.nof_re * qm_of(x.tb.mod), nullptr, &p.front, data[i], false};
  }
  // the front end of all grants at once (n > 1)
  bool                     front_done = false;
  size_t                   job_cur = 0, tile_cur = 0;
  const sch::GroupFrontEnd group   = [&](hipStream_t st, const uint32_t* which, void* const* d_e, uint32_t m) -> bool {
    if (!front_done) {
      auto* ej = reinterpret_cast<modem::EqJob*>(s.pin + o_eqj);
      for (uint32_t k = 0; k < n; k++) {
        const srsran_hip_pusch_rx_t& x = g[ord[k]];
        ej[k] = {(uint32_t)((pl[ord[k]].o_sym + (size_t)x.tb.nof_re * sizeof(cf_t)) / (2 * sizeof(cf_t))), x.noise_estimate, x.noise_estimate > 0.f ? 1u : 0u};
      }
      if (modem::launch_eq_jobs(s.pin, s.pin + region, s.dev, ej, n, (uint32_t)(tot / (2 * sizeof(cf_t))), 1.0f, st) != hipSuccess) {
        set_error("grant front end: equaliser launch failed");
        return false;
      }
      for (uint32_t k = 0; k < n;) { // one transform launch per run of grants of one allocation size
        uint32_t e = k, symb = 0;
        while (e < n && g[ord[e]].L_prb == g[ord[k]].L_prb) {
          symb += pl[ord[e]].nof_symb;
          e++;
        }
        const PuschPlan& p0 = pl[ord[k]];
        if (srsran_hip_dft_batch_run(p0.plan, (const cf_t*)(s.dev + p0.o_x), (cf_t*)(s.dev + p0.o_z), symb, st) != SRSRAN_SUCCESS) {
          return false;
        }
        k = e;
      }
      front_done = true;
    }
    const bool    llr8 = g[which[0]].tb.llr_is_8bit != 0;
    const size_t  es   = llr8 ? 1 : 2;
    modem::Params p;
    if (!modem::params_for(p, llr8 ? modem::LLR_I8 : modem::LLR_I16)) {
      return false;
    }
    uint8_t* base = static_cast<uint8_t*>(d_e[0]);
    for (uint32_t k = 1; k < m; k++) {
      base = static_cast<uint8_t*>(d_e[k]) < base ? static_cast<uint8_t*>(d_e[k]) : base;
    }
    auto*    mj  = reinterpret_cast<modem::Job*>(s.pin + o_mj) + job_cur;
    auto*    tj  = reinterpret_cast<uint32_t*>(s.pin + o_tj) + tile_cur;
    uint32_t nt  = 0;
    for (uint32_t k = 0; k < m; k++) {
      const srsran_hip_pusch_rx_t& x  = g[which[k]];
      const uint32_t               t0 = nt, cnt = modem::tiles_of(x.tb.mod, x.tb.nof_re);
      mj[k] = modem::Job{x.tb.mod, x.tb.nof_re, (uint32_t)(pl[which[k]].o_x / sizeof(cf_t)), (uint32_t)((size_t)(static_cast<uint8_t*>(d_e[k]) - base) / es), x.tb.seed, 1u, t0, cnt,
                         12 * x.L_prb, pl[which[k]].nof_symb};
      for (uint32_t t = 0; t < cnt; t++) {
        tj[nt++] = k;
      }
    }
    job_cur += m;
    tile_cur += nt;
    p.in       = s.dev + region;
    p.out      = base;
    p.jobs     = mj;
    p.tile_job = tj;
    p.n_jobs   = m;
    p.n_tiles  = nt;
    if (modem::launch(p, st) != hipSuccess) {
      set_error("grant front end: demodulator launch failed");
      return false;
    }
    return true;
  };
  sch::decode_tbs_staged(items.data(), n, n > 1 ? &group : nullptr);
  for (uint32_t i = 0; i < n; i++) {
    res[i].crc_ok               = items[i].ok ? 1 : 0;
    res[i].avg_iterations_block = pl[i].head.avg_iterations;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_pusch_decode(const srsran_hip_pusch_rx_t* g, const cf_t* sf_symbols, const cf_t* ce, srsran_softbuffer_rx_t* softbuffer,
                                       uint8_t* data, srsran_hip_grant_res_t* res)
{
  return srsran_hip_pusch_decode_multi(1, g, &sf_symbols, &ce, &softbuffer, &data, res);
}

// ------------------------------------------------------------------------------------------------ PDSCH receive, one codeword

extern "C" int srsran_hip_pdsch_decode(const srsran_hip_pdsch_rx_t* g, const cf_t* symbols, const cf_t* ce, srsran_softbuffer_rx_t* softbuffer,
                                       uint8_t* data, srsran_hip_grant_res_t* res)
{
  return srsran_hip_pdsch_decode_dbg(g, symbols, ce, softbuffer, data, res, nullptr, nullptr);
}

extern "C" int srsran_hip_pdsch_decode_dbg(const srsran_hip_pdsch_rx_t* g, const cf_t* symbols, const cf_t* ce, srsran_softbuffer_rx_t* softbuffer,
                                           uint8_t* data, srsran_hip_grant_res_t* res, cf_t* d_out, void* e_out)
{
  TraceRange trace_("srsran_hip_pdsch_decode");
  if (!g || !symbols || !softbuffer || !data || !res) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *res = {0, 0.f, NAN};
  if (!tb_valid(g->tb, "srsran_hip_pdsch_decode") || (ce && !(g->scaling != 0.f))) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_pdsch_decode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  bind_thread();
  ChanStage&      s = stage();
  srsran_cbsegm_t seg;
  if (srsran_cbsegm(&seg, g->tb.tbs) != SRSRAN_SUCCESS) {
    fprintf(stderr, "Error computing segmentation for TBS=%d\n", g->tb.tbs);
    return SRSRAN_ERROR;
  }
  const size_t nb = al256((size_t)g->tb.nof_re * sizeof(cf_t));
  const size_t ne = (size_t)g->tb.nof_re * qm_of(g->tb.mod) * (g->tb.llr_is_8bit ? 1 : 2);
  if (!s.grow(3 * nb + al256(ne), nb)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_pdsch_decode: staging allocation failed\n");
    return SRSRAN_ERROR;
  }
  memcpy(s.pin, symbols, (size_t)g->tb.nof_re * sizeof(cf_t));
  if (ce) {
    memcpy(s.pin + nb, ce, (size_t)g->tb.nof_re * sizeof(cf_t));
  }
  srsran_hip_sch_head_t       head = {g->tb.max_nof_iterations, 0.f, g->tb.llr_is_8bit != 0};
  const srsran_hip_grant_tb_t tb   = g->tb;
  const uint8_t *             psym = s.pin, *pce = ce ? s.pin + nb : nullptr;
  uint8_t*                    dx = s.dev;
  const float                 scaling = g->scaling, noise = g->noise_estimate;
  uint8_t *                   p_d = s.pin + 2 * nb, *p_e = s.pin + 3 * nb;
  const bool                  want_d = d_out && ce, want_e = e_out != nullptr;
  const sch::FrontEnd front = [=](hipStream_t st, void* d_e) {
    if (!enqueue_rx_front(st, tb, psym, pce, scaling, noise, 0, 0, dx, nullptr, nullptr, d_e)) {
      return false;
    }
    // what the reference leaves in q->d / q->e for its callers to look at
    if ((want_d && hipMemcpyAsync(p_d, dx, (size_t)tb.nof_re * sizeof(cf_t), hipMemcpyDeviceToHost, st) != hipSuccess) ||
        (want_e && hipMemcpyAsync(p_e, d_e, ne, hipMemcpyDeviceToHost, st) != hipSuccess)) {
      set_error("grant front end: copy of the intermediate results failed");
      return false;
    }
    return true;
  };
  const bool ok = sch::decode_tb_staged(&head, softbuffer, &seg, qm_rm(tb), tb.rv, tb.nof_re * qm_of(tb.mod), nullptr, &front, data);
  if (want_d) {
    memcpy(d_out, p_d, (size_t)tb.nof_re * sizeof(cf_t));
  }
  if (want_e) {
    memcpy(e_out, p_e, ne);
  }
  res->crc_ok               = ok ? 1 : 0;
  res->avg_iterations_block = head.avg_iterations;
  return SRSRAN_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ transmit side

namespace {
bool enqueue_mod(hipStream_t st, const uint8_t* d_bits, uint32_t mod, uint32_t n, uint32_t seed, bool scramble, float scale, uint8_t* p_out)
{
  modem::Params sp;
  if (!modem::params_for(sp, modem::LLR_I16)) {
    return false;
  }
  const float2* tab = modem::mod_tables();
  if (!tab) {
    return false;
  }
  modem::ModParams p = {};
  p.bits     = d_bits;
  p.out      = reinterpret_cast<float2*>(p_out);
  p.table    = tab;
  p.mod      = mod;
  p.n        = n;
  p.seed     = seed;
  p.scramble = scramble ? 1u : 0u;
  p.scale    = scale;
  p.x1_bits  = sp.x1_bits;
  p.x2_cols  = sp.x2_cols;
  if (modem::launch_mod(p, st) != hipSuccess) {
    set_error("modulator launch failed");
    return false;
  }
  return true;
}
} // namespace

extern "C" int srsran_hip_pdsch_encode(const srsran_hip_pdsch_tx_t* g, srsran_softbuffer_tx_t* softbuffer, uint8_t* data, cf_t* symbols)
{
  return srsran_hip_pdsch_encode_dbg(g, softbuffer, data, symbols, nullptr);
}

extern "C" int srsran_hip_pdsch_encode_dbg(const srsran_hip_pdsch_tx_t* g, srsran_softbuffer_tx_t* softbuffer, uint8_t* data, cf_t* symbols, uint8_t* e_out)
{
  TraceRange trace_("srsran_hip_pdsch_encode");
  if (!g || !softbuffer || !symbols) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!tb_valid(g->tb, "srsran_hip_pdsch_encode")) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_pdsch_encode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  bind_thread();
  ChanStage&      s = stage();
  srsran_cbsegm_t seg;
  if (srsran_cbsegm(&seg, g->tb.tbs) != SRSRAN_SUCCESS) {
    fprintf(stderr, "Error computing segmentation for TBS=%d\n", g->tb.tbs); // sch.c:637-640
    return SRSRAN_ERROR;
  }
  const size_t nb = (size_t)g->tb.nof_re * sizeof(cf_t);
  const size_t nw = (((size_t)g->tb.nof_re * qm_of(g->tb.mod) + 31) / 32) * 4; // scrambled bits, whole words
  if (!s.grow(al256(nb) + al256(nw), al256(nw))) {
    return SRSRAN_ERROR;
  }
  const srsran_hip_grant_tb_t tb    = g->tb;
  uint8_t *                   pout = s.pin, *p_e = s.pin + al256(nb), *d_scr = s.dev;
  const float                 scale = g->scaling != 0.f ? g->scaling : 1.0f;
  const bool                  want_e = e_out != nullptr;
  const sch::BackEnd          back  = [=](hipStream_t st, const uint8_t* d_e) {
    if (!want_e) {
      return enqueue_mod(st, d_e, tb.mod, tb.nof_re, tb.seed, true, scale, pout);
    }
    // the scrambled bits are wanted by themselves (q->e): scramble in packed form, hand that to the modulator and to the host
    modem::Params sp;
    if (!modem::params_for(sp, modem::LLR_I16)) {
      return false;
    }
    if (modem::launch_scramble_packed(d_e, d_scr, tb.nof_re * qm_of(tb.mod), tb.seed, sp.x1_bits, sp.x2_cols, st) != hipSuccess ||
        hipMemcpyAsync(p_e, d_scr, nw, hipMemcpyDeviceToHost, st) != hipSuccess) {
      set_error("packed scrambler launch failed");
      return false;
    }
    return enqueue_mod(st, d_scr, tb.mod, tb.nof_re, 0, false, scale, pout);
  };
  const int rc = sch::encode_tb_staged(softbuffer, &seg, qm_rm(tb), tb.rv, tb.nof_re * qm_of(tb.mod), data, nullptr, &back);
  if (rc != SRSRAN_SUCCESS) {
    return rc;
  }
  memcpy(symbols, s.pin, nb);
  if (want_e) {
    memcpy(e_out, p_e, ((size_t)tb.nof_re * qm_of(tb.mod) + 7) / 8);
  }
  return SRSRAN_SUCCESS;
}

// the codewords of a TTI (srsenb/src/phy/lte/cc_worker.cc encode_pdsch: one srsran_enb_dl_put_pdsch per scheduled UE) in ONE call: one coding launch over
// the code blocks of all of them, one scrambling + modulation launch, one host wait
extern "C" int srsran_hip_pdsch_encode_multi(uint32_t n, const srsran_hip_pdsch_tx_t* g, srsran_softbuffer_tx_t* const* softbuffers, uint8_t* const* data,
                                             cf_t* const* symbols)
{
  TraceRange trace_("srsran_hip_pdsch_encode");
  if (n == 0) {
    return SRSRAN_SUCCESS;
  }
  if (!g || !softbuffers || !data || !symbols) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (n == 1) {
    return srsran_hip_pdsch_encode(&g[0], softbuffers[0], data[0], symbols[0]);
  }
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_pdsch_encode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  bind_thread();
  ChanStage&                   s = stage();
  std::vector<srsran_cbsegm_t> seg(n);
  std::vector<sch::TxItem>     items(n);
  std::vector<size_t>          o_out(n);
  size_t                       out_bytes = 0, tiles = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (!softbuffers[i] || !symbols[i] || !tb_valid(g[i].tb, "srsran_hip_pdsch_encode")) {
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    if (srsran_cbsegm(&seg[i], g[i].tb.tbs) != SRSRAN_SUCCESS) {
      fprintf(stderr, "Error computing segmentation for TBS=%d\n", g[i].tb.tbs); // sch.c:637-640
      return SRSRAN_ERROR;
    }
    items[i] = {softbuffers[i], &seg[i], qm_rm(g[i].tb), g[i].tb.rv, g[i].tb.nof_re * qm_of(g[i].tb.mod), data[i], 0};
    o_out[i] = out_bytes;
    out_bytes += al256((size_t)g[i].tb.nof_re * sizeof(cf_t));
    tiles += (g[i].tb.nof_re + MODEM_TILE_SYMS - 1) / MODEM_TILE_SYMS;
  }
  const size_t o_jobs = out_bytes, o_tj = al256(o_jobs + n * sizeof(modem::ModJob));
  if (!s.grow(al256(o_tj + tiles * sizeof(uint32_t)), 0)) {
    return SRSRAN_ERROR;
  }
  const sch::GroupBackEnd back = [&](hipStream_t st, const uint8_t* d_e, const uint32_t* e_byte_off, uint32_t m) -> bool {
    modem::Params sp;
    const float2* tab = modem::mod_tables();
    if (m != n || !modem::params_for(sp, modem::LLR_I16) || !tab) {
      return false;
    }
    auto*    mj = reinterpret_cast<modem::ModJob*>(s.pin + o_jobs);
    auto*    tj = reinterpret_cast<uint32_t*>(s.pin + o_tj);
    uint32_t nt = 0;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t cnt = (g[i].tb.nof_re + MODEM_TILE_SYMS - 1) / MODEM_TILE_SYMS;
      mj[i] = {g[i].tb.mod, g[i].tb.nof_re, g[i].tb.seed, 1u, g[i].scaling != 0.f ? g[i].scaling : 1.0f, e_byte_off[i], (uint32_t)(o_out[i] / sizeof(cf_t)), nt};
      for (uint32_t t = 0; t < cnt; t++) {
        tj[nt++] = i;
      }
    }
    modem::ModParams p = {};
    p.bits     = d_e;
    p.out      = reinterpret_cast<float2*>(s.pin);
    p.table    = tab;
    p.x1_bits  = sp.x1_bits;
    p.x2_cols  = sp.x2_cols;
    p.jobs     = mj;
    p.tile_job = tj;
    if (modem::launch_mod_jobs(p, nt, st) != hipSuccess) {
      set_error("modulator launch failed");
      return false;
    }
    return true;
  };
  const int rc = sch::encode_tbs_staged(items.data(), n, &back);
  if (rc != SRSRAN_SUCCESS) {
    return rc;
  }
  for (uint32_t i = 0; i < n; i++) {
    memcpy(symbols[i], s.pin + o_out[i], (size_t)g[i].tb.nof_re * sizeof(cf_t));
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_ulsch_encode(const srsran_hip_grant_tb_t* tbp, uint32_t nof_symb, srsran_softbuffer_tx_t* softbuffer, uint8_t* data, uint8_t* q_bits)
{
  TraceRange trace_("srsran_hip_ulsch_encode");
  if (!tbp || !softbuffer || !q_bits) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  const srsran_hip_grant_tb_t tb = *tbp;
  if (!tb_valid(tb, "srsran_hip_ulsch_encode") || nof_symb == 0 || tb.nof_re % nof_symb || tb.mod < SRSRAN_MOD_QPSK || tb.mod > SRSRAN_MOD_64QAM) {
    fprintf(stderr, "Invalid input\n"); // sch.c:1213-1221
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_ulsch_encode: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  bind_thread();
  ChanStage&      s = stage();
  srsran_cbsegm_t seg;
  if (srsran_cbsegm(&seg, tb.tbs) != SRSRAN_SUCCESS) {
    fprintf(stderr, "Error computing segmentation for TBS=%d\n", tb.tbs); // sch.c:1224-1228
    return SRSRAN_ERROR;
  }
  const uint32_t Qm = qm_of(tb.mod);
  const size_t   nb = ((size_t)tb.nof_re * Qm + 7) / 8;
  if (!s.grow(al256(nb), 0)) {
    return SRSRAN_ERROR;
  }
  uint8_t*           pout = s.pin;
  const sch::BackEnd back = [=](hipStream_t st, const uint8_t* d_e) {
    if (chan::launch_ul_interleave_bits(d_e, pout, tb.nof_re, Qm, nof_symb, st) != hipSuccess) {
      set_error("channel interleaver launch failed");
      return false;
    }
    return true;
  };
  const int rc = sch::encode_tb_staged(softbuffer, &seg, Qm, tb.rv, tb.nof_re * Qm, data, nullptr, &back);
  if (rc != SRSRAN_SUCCESS) {
    return rc;
  }
  memcpy(q_bits, s.pin, nb);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_modulate_bytes(uint32_t mod, const uint8_t* bits, cf_t* symbols, uint32_t nbits, uint32_t seed, uint32_t scramble, float scaling)
{
  if (mod > SRSRAN_MOD_256QAM || !bits || !symbols) {
    return -1;
  }
  const uint32_t Qm = qm_of(mod);
  if (nbits % Qm) {
    fprintf(stderr, "Error modulator expects number of bits (%d) to be multiple of %d\n", nbits, Qm); // mod.c:141-144
    return -1;
  }
  const uint32_t n = nbits / Qm;
  if (n == 0) {
    return 0;
  }
  if (scramble && nbits > SRSRAN_HIP_SEQUENCE_MAX_LEN) {
    return -1;
  }
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_hip_modulate_bytes: %s (there is no CPU fallback)\n", get_error());
    return -1;
  }
  bind_thread();
  hipStream_t st = sch::stage_stream();
  ChanStage&  s  = stage();
  const size_t o_out = al256((nbits + 7) / 8 + 1);
  if (!st || !s.grow(o_out + (size_t)n * sizeof(cf_t), 0)) {
    return -1;
  }
  memcpy(s.pin, bits, (nbits + 7) / 8);
  if (!enqueue_mod(st, s.pin, mod, n, seed, scramble != 0, scaling != 0.f ? scaling : 1.0f, s.pin + o_out)) {
    (void)hipStreamSynchronize(st);
    return -1;
  }
  if (hipStreamSynchronize(st) != hipSuccess) {
    return -1;
  }
  memcpy(symbols, s.pin + o_out, (size_t)n * sizeof(cf_t));
  return (int)n;
}

// ------------------------------------------------------------------------------------------------ warm start
//
// The first grant of a process used to cost 20-28 ms (profiles/r03_ref_programs.json: pdsch_test -X 1): the device code of every kernel on the path is
// loaded at its first launch, the thread's staging contexts create their stream, pinned and device images, decoder and encoder objects, transform plans,
// and every (block size, redundancy version) brings its rate-matching table.  The reference does that kind of work in srsran_sch_init (sch.c:159-197:
// allocation, srsran_tdec_init, srsran_rm_turbo_gentables) -- so does the library: srsran_rm_turbo_gentables() builds every rate-matching table in
// one allocation and warms ONE worker's contexts; srsran_hip_warmup(n) makes that n.  A warm context is made by running real calls -- the largest
// grant of a 100-PRB cell, a one-block grant and a scalar-decoder grant, receive and transmit side, 16- and 8-bit soft bits -- on a short-lived
// thread whose contexts go back to the pools (hip_common.h: StagePool) when it ends.

#include <atomic>
#include <condition_variable>
#include <thread>

#include "turbo_device.h"
namespace phyhip {
namespace rm {
bool build_all_tables(); // rm_host.cpp
}
} // namespace phyhip

namespace {

struct HostSoftbuffers {
  std::vector<std::vector<int16_t>> rows;
  std::vector<std::vector<uint8_t>> keep, txrows;
  std::vector<int16_t*>             rp;
  std::vector<uint8_t*>             kp, tp;
  std::vector<uint8_t>              flags; // bool-sized
  srsran_softbuffer_rx_t            rx;
  srsran_softbuffer_tx_t            tx;
  explicit HostSoftbuffers(uint32_t n) : rows(n), keep(n), txrows(n), rp(n), kp(n), tp(n), flags(n, 0)
  {
    for (uint32_t i = 0; i < n; i++) {
      rows[i].assign(SRSRAN_HIP_SOFTBUFFER_CB_SIZE, 0);
      keep[i].assign(SRSRAN_HIP_SOFTBUFFER_CB_SIZE / 8, 0);
      txrows[i].assign(SRSRAN_HIP_SOFTBUFFER_CB_SIZE, 0);
      rp[i] = rows[i].data();
      kp[i] = keep[i].data();
      tp[i] = txrows[i].data();
    }
    rx = {n, SRSRAN_HIP_SOFTBUFFER_CB_SIZE, rp.data(), kp.data(), reinterpret_cast<bool*>(flags.data()), false};
    tx = {n, SRSRAN_HIP_SOFTBUFFER_CB_SIZE, tp.data()};
  }
  void reset()
  {
    for (auto& r : rows) {
      std::fill(r.begin(), r.end(), 0);
    }
    std::fill(flags.begin(), flags.end(), 0);
  }
};

void warm_one_worker()
{
  const uint32_t nof_prb = 100, L_prb = 100, nsymb = 12;
  // transport block sizes without filler bits: C blocks of K = 6144 carry C (6144 - 24) - 24 payload bits (C > 1), one block K - 24
  const struct {
    uint32_t tbs, mod, L;
  } grants[] = {{13 * 6120 - 24, SRSRAN_MOD_64QAM, L_prb}, {6144 - 24, SRSRAN_MOD_16QAM, 12}, {40 - 24, SRSRAN_MOD_QPSK, 1}};
  HostSoftbuffers      sb(13);
  std::vector<cf_t>    grid((size_t)14 * 12 * nof_prb, cf_t(0.5f, -0.5f)), ce((size_t)14 * 12 * nof_prb, cf_t(1.f, 0.f)), sym((size_t)nsymb * 12 * L_prb);
  std::vector<uint8_t> data(13 * 768 + 64, 0x5a), qbits((size_t)nsymb * 12 * L_prb * 6 / 8 + 8);
  for (uint32_t llr8 = 0; llr8 < 2; llr8++) {
    for (const auto& gr : grants) {
      const uint32_t        nof_re = nsymb * 12 * gr.L;
      srsran_hip_grant_tb_t tb     = {gr.mod, gr.tbs, 0, nof_re, 12345u, 1, llr8, 1};
      srsran_hip_grant_res_t res;
      // receive: PUSCH grant from the grid, PDSCH codeword with and without the equaliser
      srsran_hip_pusch_rx_t pu = {tb, nof_prb, 7, {0, 0}, gr.L, 0, 0.01f, 0};
      sb.reset();
      (void)srsran_hip_pusch_decode(&pu, grid.data(), ce.data(), &sb.rx, data.data(), &res);
      srsran_hip_pdsch_rx_t pd = {tb, 1.0f, 0.01f};
      sb.reset();
      (void)srsran_hip_pdsch_decode_dbg(&pd, grid.data(), ce.data(), &sb.rx, data.data(), &res, sym.data(), nullptr);
      // (a retransmission: the rows that came back are combined into)
      tb.rv = 2;
      pd.tb = tb;
      (void)srsran_hip_pdsch_decode(&pd, grid.data(), nullptr, &sb.rx, data.data(), &res);
      tb.rv = 0;
      if (!llr8) { // transmit
        srsran_hip_pdsch_tx_t tx = {tb, 1.0f};
        (void)srsran_hip_pdsch_encode_dbg(&tx, &sb.tx, data.data(), sym.data(), qbits.data());
        (void)srsran_hip_ulsch_encode(&tb, nsymb, &sb.tx, data.data(), qbits.data());
      }
    }
  }
}

struct WarmState { // per device
  std::mutex mu;
  uint32_t   workers = 0;
};

} // namespace

extern "C" int srsran_hip_warmup(uint32_t nof_workers)
{
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  bind_thread();
  WarmState&                  ws = device_local<WarmState>(); // of the calling thread's device
  std::lock_guard<std::mutex> lk(ws.mu);
  const int                   dev = current_device();
  if (!rm::build_all_tables() || !turbo::prebuild_tables()) {
    return SRSRAN_ERROR;
  }
  // the workers' contexts are made by threads that exist TOGETHER (a context goes back to the pool when its thread ends: one after the other they
  // would all warm the same one)
  const uint32_t have = ws.workers;
  if (have < nof_workers) {
    const uint32_t           n = nof_workers - have;
    std::mutex               mu;
    std::condition_variable  cv;
    uint32_t                 done = 0;
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n; i++) {
      th.emplace_back([&] {
        (void)srsran_hip_set_thread_device(dev);
        warm_one_worker();
        std::unique_lock<std::mutex> l(mu);
        done++;
        cv.notify_all();
        cv.wait(l, [&] { return done == n; });
      });
    }
    for (auto& t : th) {
      t.join(); // their contexts are in the pools now
    }
    ws.workers = nof_workers;
  }
  return SRSRAN_SUCCESS;
}
