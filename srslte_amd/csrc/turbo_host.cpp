// turbo_host.cpp -- host side of the turbo decoder: QPP tables, batch object, srsran_tdec_* handle ABI.
//
// Mirrors (interface + behaviour, not code) lib/src/phy/fec/turbo/turbodecoder.c, tc_interl_lte.c,
// tc_interl_umts.c:51-90 (init/free) and lib/src/phy/fec/cbsegm.c:119-140 of the reference.
#include "coalesce.h"
#include "hip_common.h"
#include "srsran_amd/phy_sch_abi.h"
#include "tables/lte_qpp_table.h"
#include "turbo_device.h"

#include <map>
#include <mutex>
#include <thread>
#include <vector>

using namespace phyhip;

// ------------------------------------------------------------------------------------------------ tables

extern "C" int srsran_cbsegm_cbindex(uint32_t long_cb)
{
  // cbsegm.c:119-130: first table entry >= long_cb
  int j = 0;
  while (j < LTE_QPP_NOF_SIZES && lte_qpp_table[j][0] < long_cb) {
    j++;
  }
  return (j == LTE_QPP_NOF_SIZES) ? SRSRAN_ERROR : j;
}

extern "C" int srsran_cbsegm_cbsize(uint32_t index)
{
  return (index < LTE_QPP_NOF_SIZES) ? (int)lte_qpp_table[index][0] : SRSRAN_ERROR;
}

extern "C" bool srsran_cbsegm_cbsize_isvalid(uint32_t size) // cbsegm.c:142-150
{
  const int idx = srsran_cbsegm_cbindex(size);
  return idx >= 0 && (uint32_t)srsran_cbsegm_cbsize((uint32_t)idx) == size;
}

extern "C" int srsran_tc_interl_init(srsran_tc_interl_t* h, uint32_t max_long_cb)
{
  h->max_long_cb = max_long_cb;
  h->forward     = (uint16_t*)calloc(max_long_cb ? max_long_cb : 1, sizeof(uint16_t));
  h->reverse     = (uint16_t*)calloc(max_long_cb ? max_long_cb : 1, sizeof(uint16_t));
  if (!h->forward || !h->reverse) {
    perror("malloc");
    srsran_tc_interl_free(h);
    return SRSRAN_ERROR;
  }
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_tc_interl_free(srsran_tc_interl_t* h)
{
  if (h->forward) {
    free(h->forward);
  }
  if (h->reverse) {
    free(h->reverse);
  }
  memset(h, 0, sizeof(*h));
}

// natural-order QPP permutation PI(i) = (f1 i + f2 i^2) mod K and its inverse
static int qpp_natural(uint32_t K, std::vector<uint16_t>& fwd, std::vector<uint16_t>& rev)
{
  int idx = srsran_cbsegm_cbindex(K);
  if (idx < 0) {
    return SRSRAN_ERROR;
  }
  const uint64_t f1 = lte_qpp_table[idx][1], f2 = lte_qpp_table[idx][2];
  fwd.resize(K);
  rev.resize(K);
  for (uint64_t i = 0; i < K; i++) {
    uint64_t j = (f1 * i + f2 * i * i) % K;
    fwd[i]     = (uint16_t)j;
    rev[j]     = (uint16_t)i;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_tc_interl_LTE_gen_interl(srsran_tc_interl_t* h, uint32_t long_cb, uint32_t interl_win)
{
  if (long_cb > h->max_long_cb) {
    fprintf(stderr, "Interleaver initiated for max_long_cb=%d\n", h->max_long_cb);
    return SRSRAN_ERROR;
  }
  std::vector<uint16_t> f, r;
  if (qpp_natural(long_cb, f, r) || interl_win == 0 || long_cb % interl_win) {
    fprintf(stderr, "Can't find long_cb=%d in valid TC CB table\n", long_cb);
    return SRSRAN_ERROR;
  }
  if (interl_win == 1) {
    memcpy(h->forward, f.data(), long_cb * sizeof(uint16_t));
    memcpy(h->reverse, r.data(), long_cb * sizeof(uint16_t));
  } else {
    // tc_interl_lte.c:90-106: the same permutation seen through the [step][sub-block] lane order
    const uint32_t sb = long_cb / interl_win;
    for (uint32_t i = 0; i < long_cb; i++) {
      uint32_t nat = (i % interl_win) * sb + i / interl_win;
      uint32_t a = f[nat], b = r[nat];
      h->forward[i] = (uint16_t)((a % sb) * interl_win + a / sb);
      h->reverse[i] = (uint16_t)((b % sb) * interl_win + b / sb);
    }
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_tc_interl_LTE_gen(srsran_tc_interl_t* h, uint32_t long_cb)
{
  return srsran_tc_interl_LTE_gen_interl(h, long_cb, 1);
}

extern "C" uint32_t srsran_tdec_autoimp_get_subblocks(uint32_t long_cb)
{
  // turbodecoder.c:381-393 on an AVX2 host.  MUST stay in step with the reference: srsran_rm_turbo_rx_lut
  // picks its output layout from this value (rm_turbo.c:412-413).
  if (!(long_cb % 16) && long_cb > 800) {
    return 16;
  } else if (!(long_cb % 8) && long_cb > 400) {
    return 8;
  }
  return 0;
}

extern "C" uint32_t srsran_tdec_autoimp_get_subblocks_8bit(uint32_t long_cb)
{
  // turbodecoder.c:410-424
  if (!(long_cb % 32) && long_cb > 2048) {
    return 32;
  } else if (!(long_cb % 16) && long_cb > 800) {
    return 16;
  } else if (!(long_cb % 8) && long_cb > 400) {
    return 8;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ batch object

struct srsran_hip_tdec_batch {
  DeviceTag tag;
  uint32_t K       = 0;
  uint32_t max_cb  = 0;
  int      nb      = 0; // 32, 16, 8 (window decoders) or 0 (scalar decoder)
  bool     arith8  = false; // 8-bit window decoder arithmetic (sse8: 16 sub-blocks, avx8: 32)
  // window decoder
  uint32_t* d_ws     = nullptr;
  uint32_t  ws_stride = 0;
  uint32_t* d_deint  = nullptr;
  uint32_t* d_inter  = nullptr;
  // scalar decoder
  short*    d_ws_gen    = nullptr;
  uint16_t* d_inter16   = nullptr;
  uint16_t* d_deinter16 = nullptr;
  // optional parity aid
  short* d_dec_llr = nullptr;
  // early stop on CRC: multipliers x^(W (nb-1-d)) mod g for the generators used so far
  std::map<uint32_t, uint32_t*> crc_mult;
  // the "persistent" launch variant's unit counter (development knob) and the CU count of the object's device
  uint32_t* d_unit_counter = nullptr;
  int       cus            = 0;
  // latency kernel (small batches): its own workspace, allocated on first use for lat_cap code blocks
  uint32_t* d_ws_lat      = nullptr;
  uint32_t  lat_cap       = 0;
  std::map<uint32_t, uint32_t*> crc_mult_lat; // per generator: one multiplier per 32-bit word of the kernel's hard-bit image
  bool      state_in_lat  = false; // the decoder state of the last launch lives in the latency kernel's workspace
  bool      state_in_gen_lat = false; // scalar decoder: ... in the latency kernel's per-block layout of d_ws_gen
  // tables borrowed from the process-wide cache (never freed by the object); workspace taken from a caller's arena at every launch
  bool              tables_cached = false;
  turbo::WsArena*   arena         = nullptr;
};

// ---- process-wide cache of the small device constants a decoder of (K, sub-blocks) needs: exchange tables, interleaver tables, CRC multipliers.
// Built on first use -- or all at once, in ONE allocation, by turbo::prebuild_tables() (the init-time warm-up): creating the decoder of a block size
// a thread has not seen yet then costs no allocation, no upload and no device-wide wait.
namespace {
struct ConstCache {
  std::mutex                mu;
  std::map<uint64_t, void*> dev;
  // bulk mode (prebuild): constants are collected in one host image and placed with one upload
  bool                                         bulk = false;
  std::thread::id                              bulk_owner; // only the prebuilding thread collects; any other thread keeps the normal path
  std::vector<uint8_t>                         img;
  std::vector<std::pair<uint64_t, size_t>>     pending; // key -> offset in img
};
ConstCache& ccache()
{
  return device_local<ConstCache>(); // the constants of the calling thread's device; never destroyed
}
enum { CK_DEINT = 1, CK_INTER, CK_INTER16, CK_DEINTER16, CK_CRC_WIN, CK_CRC_LAT, CK_CRC_GEN };
inline uint64_t ckey(int kind, uint32_t K, int nb, uint32_t poly)
{
  return ((uint64_t)kind << 56) | ((uint64_t)K << 40) | ((uint64_t)(uint32_t)nb << 32) | poly;
}
// device copy of the constant `key`, made from fill() on first use; nullptr on failure (in bulk mode: a non-null placeholder, valid after the upload)
template <class Fill>
void* cached_const(uint64_t key, size_t bytes, Fill fill)
{
  ConstCache&                 c = ccache();
  std::lock_guard<std::mutex> lk(c.mu);
  auto                        it = c.dev.find(key);
  if (it != c.dev.end()) {
    return it->second;
  }
  if (c.bulk && c.bulk_owner == std::this_thread::get_id()) {
    for (auto& pd : c.pending) {
      if (pd.first == key) {
        return &c; // placeholder
      }
    }
    const size_t off = (c.img.size() + 255) & ~(size_t)255;
    c.img.resize(off + bytes);
    fill(c.img.data() + off);
    c.pending.emplace_back(key, off);
    return &c;
  }
  std::vector<uint8_t> host(bytes);
  fill(host.data());
  void* d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess || upload(d, host.data(), bytes) != hipSuccess) {
    (void)hipFree(d);
    set_error("turbo decoder: cannot place a table on the device");
    return nullptr;
  }
  c.dev[key] = d;
  return d;
}
} // namespace

// which kernel takes a launch of n_cb blocks starting at half iteration n_begin (a resumed run stays where its state is)
static bool want_lat(srsran_hip_tdec_batch* h, uint32_t n_cb, uint32_t n_begin)
{
  if (!h->nb || !turbo::lat_exists(h->nb, h->arith8)) {
    return false;
  }
  if (n_begin > 0) {
    return h->state_in_lat;
  }
  const int k = knob(KNOB_TDEC_LAT);
  return k == 0 ? false : (k > 0 ? true : turbo::lat_waves(h->nb, n_cb) <= turbo::kLatMaxBlocks);
}
static uint32_t xpow_mod(uint64_t e, uint32_t poly);
// the same for the scalar decoder (K <= 400 with AUTO): 8 lanes per block with the block in LDS, or one lane per block
static bool want_gen_lat(srsran_hip_tdec_batch* h, uint32_t n_cb, uint32_t n_begin)
{
  if (h->nb || (h->K & 7u) || turbo::gen_lat_lds_bytes(h->K) > 156 * 1024) {
    return false;
  }
  if (n_begin > 0) {
    return h->state_in_gen_lat;
  }
  const int k = knob(KNOB_TDEC_LAT);
  return k == 0 ? false : (k > 0 ? true : n_cb <= turbo::kGenLatMaxBlocks);
}
// lane l of a block's 8 takes the bits [l c, min((l + 1) c, K)), c = ceil(K / 8): x^(K - end) mod g shifts its remainder into place
static uint32_t* gen_crc_mult8(uint32_t K, uint32_t poly)
{
  return static_cast<uint32_t*>(cached_const(ckey(CK_CRC_GEN, K, 0, poly), 8 * sizeof(uint32_t), [&](void* dst) {
    uint32_t*      m = static_cast<uint32_t*>(dst);
    const uint32_t c = (K + 7) / 8;
    for (uint32_t l = 0; l < 8; l++) {
      const uint32_t lo = l * c < K ? l * c : K, hi = lo + c < K ? lo + c : K;
      m[l]              = xpow_mod((uint64_t)K - hi, poly);
    }
  }));
}
// the latency kernel's two-wave form: 16 sub-blocks, few blocks (it takes two waves and up to 96 KB of LDS per block)
static bool want_lat2(srsran_hip_tdec_batch* h, uint32_t n_cb)
{
  if (h->nb != 16 || turbo::lat2_lds_bytes(h->K) > 120 * 1024) {
    return false;
  }
  const int k = knob(KNOB_TDEC_LAT2);
  // one round of the chip: a block's two waves and its filed rows -- up to four blocks per CU where the rows are small (K = 1024: 1024 blocks 0.155 against
  // 0.199 ms per 8 half iterations, 2048 blocks 0.37 against 0.28; K = 6144, one block per CU: 256 blocks 0.27 against 0.51, 512 blocks 0.54 against 0.57)
  const size_t per_cu = std::min<size_t>(4, (150 * 1024) / turbo::lat2_lds_bytes(h->K));
  return k == 0 ? false : (k > 0 ? true : n_cb <= turbo::kLat2MaxBlocks * per_cu);
}
static int ensure_lat_ws(srsran_hip_tdec_batch* h, uint32_t n_cb)
{
  if (n_cb <= h->lat_cap) {
    return SRSRAN_SUCCESS;
  }
  (void)hipFree(h->d_ws_lat);
  h->d_ws_lat = nullptr;
  h->lat_cap  = 0;
  const uint32_t cap = ((n_cb > turbo::kLatMaxBlocks ? n_cb : (h->max_cb < turbo::kLatMaxBlocks ? h->max_cb : turbo::kLatMaxBlocks)) + 1u) & ~1u;
  PHY_HIP_CHECK(hipMalloc(&h->d_ws_lat, (size_t)turbo::lat_ws_dwords(h->K, h->nb) * cap * sizeof(uint32_t)), SRSRAN_ERROR);
  h->lat_cap = cap;
  return SRSRAN_SUCCESS;
}

// which decoder the reference runs: sub-block count and arithmetic (turbodecoder.c:381-441,455-512)
static int impl_to_cfg(int impl, bool llr8_api, uint32_t K, int* nb, bool* arith8)
{
  *arith8 = false;
  switch (impl) {
    case SRSRAN_TDEC_AUTO:
      if (llr8_api) {
        *nb = (int)srsran_tdec_autoimp_get_subblocks_8bit(K);
        if (*nb >= 16) {
          *arith8 = true; // else: no 8-bit decoder takes this K, the LLRs are widened for sse16 / gen
        }
      } else {
        *nb = (int)srsran_tdec_autoimp_get_subblocks(K);
      }
      return 0;
    case SRSRAN_TDEC_GENERIC:
      *nb = 0;
      return 0;
    case SRSRAN_TDEC_SSE_WINDOW:
      *nb = 8;
      return 0;
    case SRSRAN_TDEC_AVX_WINDOW:
      *nb = 16;
      return 0;
    case SRSRAN_TDEC_SSE8_WINDOW:
      *nb     = 16;
      *arith8 = true;
      return 0;
    case SRSRAN_TDEC_AVX8_WINDOW:
      *nb     = 32;
      *arith8 = true;
      return 0;
    default:
      return -1;
  }
}

static int tdec_batch_create(srsran_hip_tdec_batch_t** hh, uint32_t long_cb, uint32_t max_nof_cb, int impl, bool llr8_api, turbo::WsArena* arena = nullptr);

extern "C" int srsran_hip_tdec_batch_create(srsran_hip_tdec_batch_t** hh, uint32_t long_cb, uint32_t max_nof_cb, int impl)
{
  return tdec_batch_create(hh, long_cb, max_nof_cb, impl, false);
}

extern "C" int srsran_hip_tdec_batch_create_8bit(srsran_hip_tdec_batch_t** hh, uint32_t long_cb, uint32_t max_nof_cb, int impl)
{
  return tdec_batch_create(hh, long_cb, max_nof_cb, impl, true);
}

static int tdec_batch_create(srsran_hip_tdec_batch_t** hh, uint32_t long_cb, uint32_t max_nof_cb, int impl, bool llr8_api, turbo::WsArena* arena)
{
  if (!hh || max_nof_cb == 0) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  int idx = srsran_cbsegm_cbindex(long_cb);
  if (idx < 0 || (uint32_t)srsran_cbsegm_cbsize(idx) != long_cb) {
    set_error("invalid turbo code block size %u", long_cb);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  int  nb     = 0;
  bool arith8 = false;
  if (impl_to_cfg(impl, llr8_api, long_cb, &nb, &arith8)) {
    set_error("turbo decoder implementation %d not supported by the HIP engine", impl);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (nb && (long_cb % nb || long_cb / nb <= 40)) {
    // the reference's window decoders read out of bounds in this case (40-step warm-up > sub-block); with
    // exactly 40 steps per sub-block its warm-up pass degenerates (turbodecoder_win.h:573,712).  AUTO never
    // selects either.
    set_error("window decoder with %d sub-blocks is invalid for K=%u", nb, long_cb);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  auto* h   = new srsran_hip_tdec_batch;
  h->K      = long_cb;
  h->max_cb = max_nof_cb;
  h->nb     = nb;
  h->arith8 = arith8;
  h->arena  = arena;
  h->tables_cached = true;
  const uint32_t K = long_cb;
  if (nb) {
    const uint32_t lpc = nb / 2, long_sb = K / nb, nblk = (long_sb + 7) / 8;
    h->ws_stride = turbo::win_ws_dwords(K, nb);
    // Exchange tables, one dword per (step k, destination lane p'), stored blocked [k/8][p'][k%8]:
    //   bits 0..15  destination row o = PI'(k) mod W  (common to all sub-blocks: QPP is contention free)
    //   bits 16..20 source sub-block whose output lands in destination sub-block 2p'
    //   bits 21..25 source sub-block whose output lands in destination sub-block 2p'+1
    // deint: app2[reverse[n]] = ext1[n]   inter: app1[forward[n]] = ext2[n]   (turbodecoder_iter.h:118,124)
    const size_t tb = (size_t)nblk * lpc * 8 * sizeof(uint32_t);
    bool         contention = false;
    auto         fill_dir = [&](int dir, void* dst) {
      std::vector<uint16_t> f, r;
      qpp_natural(K, f, r);
      const std::vector<uint16_t>& tab = dir == 0 ? r : f;
      uint32_t*                    out = static_cast<uint32_t*>(dst);
      memset(out, 0, tb);
      for (uint32_t k = 0; k < long_sb; k++) {
        uint32_t src_of[32];
        uint32_t row = tab[k] % long_sb;
        for (uint32_t j = 0; j < (uint32_t)nb; j++) {
          uint32_t t = tab[j * long_sb + k];
          if (t % long_sb != row) {
            contention = true;
          }
          src_of[t / long_sb] = j;
        }
        for (uint32_t pp = 0; pp < lpc; pp++) {
          out[((k >> 3) * lpc + pp) * 8 + (k & 7)] = row | (src_of[2 * pp] << 16) | (src_of[2 * pp + 1] << 21);
        }
      }
    };
    h->d_deint = static_cast<uint32_t*>(cached_const(ckey(CK_DEINT, K, nb, 0), tb, [&](void* d) { fill_dir(0, d); }));
    h->d_inter = static_cast<uint32_t*>(cached_const(ckey(CK_INTER, K, nb, 0), tb, [&](void* d) { fill_dir(1, d); }));
    if (contention) {
      set_error("QPP interleaver of K=%u is not contention free for %d windows", K, nb);
      delete h;
      return SRSRAN_ERROR;
    }
    if (!h->d_deint || !h->d_inter) {
      delete h;
      return SRSRAN_ERROR;
    }
    if (!arena) {
      // one slab per wave (64/lpc code blocks); round the block count up to whole waves
      const uint32_t cpw = 64 / lpc;
      PHY_HIP_CHECK(hipMalloc(&h->d_ws, (size_t)h->ws_stride * ceil_div(max_nof_cb, cpw) * cpw * sizeof(uint32_t)), SRSRAN_ERROR);
    }
  } else {
    auto fill16 = [&](int dir, void* dst) {
      std::vector<uint16_t> f, r;
      qpp_natural(K, f, r);
      memcpy(dst, (dir == 0 ? f : r).data(), K * sizeof(uint16_t));
    };
    h->d_inter16   = static_cast<uint16_t*>(cached_const(ckey(CK_INTER16, K, 0, 0), K * sizeof(uint16_t), [&](void* d) { fill16(0, d); }));
    h->d_deinter16 = static_cast<uint16_t*>(cached_const(ckey(CK_DEINTER16, K, 0, 0), K * sizeof(uint16_t), [&](void* d) { fill16(1, d); }));
    if (!h->d_inter16 || !h->d_deinter16) {
      delete h;
      return SRSRAN_ERROR;
    }
    if (!arena) {
      size_t nwaves = ceil_div(max_nof_cb, 64);
      PHY_HIP_CHECK(hipMalloc(&h->d_ws_gen, turbo::gen_ws_shorts(K) * nwaves * sizeof(short)), SRSRAN_ERROR);
    }
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_tdec_batch_free(srsran_hip_tdec_batch_t* h)
{
  if (!h) {
    return;
  }
  if (!h->arena) {
    hipFree(h->d_ws);
    hipFree(h->d_ws_lat);
    hipFree(h->d_ws_gen);
  }
  hipFree(h->d_dec_llr);
  // (exchange / interleaver tables and CRC multipliers belong to the process-wide cache)
  delete h;
}

// run half iterations [n_begin, n_end) and take the hard decision for n_iter = n_end
static int tdec_batch_run_range(srsran_hip_tdec_batch_t* h, const void* d_input_v, bool in_is8, uint32_t in_stride,
                                uint8_t* d_output, uint32_t out_stride, uint32_t n_cb, uint32_t n_begin, uint32_t n_end,
                                int sb_layout, bool want_llr, hipStream_t stream)
{
  TraceRange trace_("srsran_hip_tdec_batch_run");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_tdec_batch_run", SRSRAN_ERROR);
  }
  const int16_t* d_input = static_cast<const int16_t*>(d_input_v);
  if (h && n_cb == 0) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!h || !d_output || (!d_input && n_begin == 0) || n_cb > h->max_cb || n_end <= n_begin) {
    set_error("tdec batch: invalid arguments (n_cb=%u max=%u)", n_cb, h ? h->max_cb : 0);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (sb_layout && !h->nb) {
    set_error("tdec batch: sub-block input layout needs a window decoder");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (h->arena) {
    set_error("tdec batch: a decoder on a shared workspace runs whole transport-block launches only");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  const uint32_t need_in = sb_layout ? 3 * (h->K + 32) + 12 : 3 * h->K + 12;
  if ((n_cb > 1 && (in_stride < need_in || out_stride < h->K / 8))) {
    set_error("tdec batch: strides too small");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (want_llr && !h->d_dec_llr) {
    PHY_HIP_CHECK(hipMalloc(&h->d_dec_llr, (size_t)h->K * h->max_cb * sizeof(short)), SRSRAN_ERROR);
  }
  if (h->nb) {
    turbo::WinParams p = {};
    p.input      = d_input;
    p.output     = d_output;
    p.dec_llr    = want_llr ? h->d_dec_llr : nullptr;
    p.ws         = h->d_ws;
    p.deint      = h->d_deint;
    p.inter      = h->d_inter;
    p.in_stride  = in_stride;
    p.out_stride = out_stride;
    p.ws_stride  = h->ws_stride;
    p.K          = h->K;
    p.n_begin    = n_begin;
    p.n_end      = n_end;
    p.n_cb       = (int)n_cb;
    p.sb_layout  = sb_layout;
    p.in_is8     = in_is8 ? 1 : 0;
    if (knob(KNOB_TDEC_EXTRACT_ONLY) > 0) {
      p.n_end = 0; // development aid: input extraction + decision only
    }
    // launch-shape alternatives kept for measurement (profiles/r02_turbo_variants.txt); the product path is variant 0
    const int variant = knob(KNOB_TDEC_VARIANT) > 0 ? knob(KNOB_TDEC_VARIANT) : 0;
    if (variant == 2 && n_begin == 0 && h->nb == 16 && !h->arith8) {
      // the unit counter belongs to the batch object (launches of different objects / streams must not share one) and lives on
      // the device the object was created on
      if (!h->d_unit_counter) {
        int dev = 0;
        PHY_HIP_CHECK(hipGetDevice(&dev), SRSRAN_ERROR);
        PHY_HIP_CHECK(hipMalloc(&h->d_unit_counter, sizeof(uint32_t)), SRSRAN_ERROR);
        PHY_HIP_CHECK(hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, dev), SRSRAN_ERROR);
      }
      PHY_HIP_CHECK(hipMemsetAsync(h->d_unit_counter, 0, sizeof(uint32_t), stream), SRSRAN_ERROR);
      p.n_units      = ceil_div(n_cb, 8);
      p.max_resident = (uint32_t)h->cus * 8u; // 4 SIMDs x 2 waves
      p.unit_counter = h->d_unit_counter;
    }
    p.variant = variant;
    if (want_lat(h, n_cb, n_begin)) {
      if (ensure_lat_ws(h, n_cb)) {
        return SRSRAN_ERROR;
      }
      p.ws        = h->d_ws_lat;
      p.ws_stride = turbo::lat_ws_dwords(h->K, h->nb);
      PHY_HIP_CHECK(want_lat2(h, n_cb) ? turbo::launch_lat2(h->arith8, p, stream) : turbo::launch_lat(h->nb, h->arith8, p, stream), SRSRAN_ERROR);
      h->state_in_lat = true;
    } else {
      PHY_HIP_CHECK(turbo::launch_win(h->nb, h->arith8, p, stream), SRSRAN_ERROR);
      h->state_in_lat = false;
    }
  } else {
    turbo::GenParams p = {};
    p.input      = d_input;
    p.output     = d_output;
    p.dec_llr    = want_llr ? h->d_dec_llr : nullptr;
    p.ws         = h->d_ws_gen;
    p.inter      = h->d_inter16;
    p.deinter    = h->d_deinter16;
    p.ws_stride  = turbo::gen_ws_shorts(h->K);
    p.in_stride  = in_stride;
    p.out_stride = out_stride;
    p.K          = h->K;
    p.n_begin    = n_begin;
    p.n_end      = n_end;
    p.n_cb       = (int)n_cb;
    p.in_is8     = in_is8 ? 1 : 0;
    if (want_gen_lat(h, n_cb, n_begin)) {
      PHY_HIP_CHECK(turbo::launch_gen_lat(p, stream), SRSRAN_ERROR);
      h->state_in_gen_lat = true;
    } else {
      PHY_HIP_CHECK(turbo::launch_gen(p, stream), SRSRAN_ERROR);
      h->state_in_gen_lat = false;
    }
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_tdec_batch_run(srsran_hip_tdec_batch_t* h, const int16_t* d_input, uint32_t in_stride,
                                         uint8_t* d_output, uint32_t out_stride, uint32_t n_cb,
                                         uint32_t nof_iterations, int sb_layout, void* stream)
{
  // turbodecoder.c:542-544 is a do/while: at least one half iteration runs
  uint32_t nit = nof_iterations ? nof_iterations : 1;
  return tdec_batch_run_range(h, d_input, false, in_stride, d_output, out_stride, n_cb, 0, nit, sb_layout, false, (hipStream_t)stream);
}

extern "C" int srsran_hip_tdec_batch_run_8bit(srsran_hip_tdec_batch_t* h, const int8_t* d_input, uint32_t in_stride,
                                              uint8_t* d_output, uint32_t out_stride, uint32_t n_cb,
                                              uint32_t nof_iterations, int sb_layout, void* stream)
{
  uint32_t nit = nof_iterations ? nof_iterations : 1;
  return tdec_batch_run_range(h, d_input, true, in_stride, d_output, out_stride, n_cb, 0, nit, sb_layout, false, (hipStream_t)stream);
}

extern "C" int srsran_hip_tdec_batch_last_llr(srsran_hip_tdec_batch_t* h, int16_t* d_llr, uint32_t n_cb, void* stream)
{
  if (!h || !h->d_dec_llr || n_cb > h->max_cb) {
    set_error("tdec batch: no decision LLRs recorded (run with the debug entry point first)");
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(d_llr, h->d_dec_llr, (size_t)h->K * n_cb * sizeof(short), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream),
                SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

// debug/parity entry point (not in the public header on purpose: tests bind it by name)
extern "C" SRSRAN_API int srsran_hip_tdec_batch_run_dbg(srsran_hip_tdec_batch_t* h, const int16_t* d_input,
                                                        uint32_t in_stride, uint8_t* d_output, uint32_t out_stride,
                                                        uint32_t n_cb, uint32_t n_begin, uint32_t n_end, int sb_layout,
                                                        void* stream)
{
  return tdec_batch_run_range(h, d_input, false, in_stride, d_output, out_stride, n_cb, n_begin, n_end, sb_layout, true,
                              (hipStream_t)stream);
}

extern "C" SRSRAN_API int srsran_hip_tdec_batch_run_dbg_8bit(srsran_hip_tdec_batch_t* h, const int8_t* d_input,
                                                             uint32_t in_stride, uint8_t* d_output, uint32_t out_stride,
                                                             uint32_t n_cb, uint32_t n_begin, uint32_t n_end, int sb_layout,
                                                             void* stream)
{
  return tdec_batch_run_range(h, d_input, true, in_stride, d_output, out_stride, n_cb, n_begin, n_end, sb_layout, true,
                              (hipStream_t)stream);
}

// x^e mod g over GF(2), g of degree 24 given with its x^24 term
static uint32_t xpow_mod(uint64_t e, uint32_t poly)
{
  auto mul = [&](uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int i = 23; i >= 0; i--) {
      r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? (poly & 0xffffffu) : 0u);
      if ((b >> i) & 1u) {
        r ^= a;
      }
    }
    return r;
  };
  uint32_t result = 1, base = 2; // the polynomials 1 and x
  while (e) {
    if (e & 1) {
      result = mul(result, base);
    }
    base = mul(base, base);
    e >>= 1;
  }
  return result;
}

// Transport-block decoding (sch_host.cpp): all half iterations up to max_iterations with the per-block CRC early
// stop of decode_tb_cb (sch.c:420-454); d_desc places every block's input / output.
int phyhip::turbo::batch_run_early_stop(srsran_hip_tdec_batch_t* h, const void* d_input, bool in_is8, const turbo::CbDesc* d_desc,
                                        uint8_t* d_output, uint32_t n_cb, uint32_t max_iterations, int sb_layout, uint32_t crc_poly,
                                        int* d_noi, uint8_t* d_crc_ok, hipStream_t stream)
{
  if (h) {
    PHY_DEV_GUARD(h->tag, "tdec early stop", SRSRAN_ERROR);
  }
  if (!h || !d_input || !d_output || !d_desc || n_cb == 0 || n_cb > h->max_cb || max_iterations == 0 || !crc_poly || (sb_layout && !h->nb)) {
    set_error("tdec early stop: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!h->nb) {
    // K <= 400: the scalar decoder (turbodecoder.c:381-408 sends these blocks to gen_impl), one lane per code block
    turbo::GenParams g = {};
    g.input     = static_cast<const short*>(d_input);
    g.output    = d_output;
    g.ws        = h->arena ? static_cast<short*>(h->arena->ensure(turbo::gen_ws_shorts(h->K) * ceil_div(n_cb, 64) * sizeof(short), stream)) : h->d_ws_gen;
    if (!g.ws) {
      return SRSRAN_ERROR;
    }
    g.inter     = h->d_inter16;
    g.deinter   = h->d_deinter16;
    g.ws_stride = turbo::gen_ws_shorts(h->K);
    g.K         = h->K;
    g.n_begin   = 0;
    g.n_end     = max_iterations;
    g.n_cb      = (int)n_cb;
    g.in_is8    = in_is8 ? 1 : 0;
    g.desc      = d_desc;
    g.crc_poly  = crc_poly;
    g.noi       = d_noi;
    g.crc_ok    = d_crc_ok;
    if (want_gen_lat(h, n_cb, 0)) {
      g.crc_mult8 = gen_crc_mult8(h->K, crc_poly);
      if (!g.crc_mult8) {
        return SRSRAN_ERROR;
      }
      PHY_HIP_CHECK(turbo::launch_gen_lat(g, stream), SRSRAN_ERROR);
      h->state_in_gen_lat = true;
      return SRSRAN_SUCCESS;
    }
    h->state_in_gen_lat = false;
    PHY_HIP_CHECK(turbo::launch_gen(g, stream), SRSRAN_ERROR);
    return SRSRAN_SUCCESS;
  }
  const uint32_t nbq = (uint32_t)h->nb, Kq = h->K;
  uint32_t*      d_mult = static_cast<uint32_t*>(cached_const(ckey(CK_CRC_WIN, Kq, h->nb, crc_poly), nbq * sizeof(uint32_t), [&](void* dst) {
    uint32_t*      m = static_cast<uint32_t*>(dst);
    const uint64_t W = Kq / nbq;
    for (uint32_t d = 0; d < nbq; d++) {
      m[d] = xpow_mod(W * (uint64_t)(nbq - 1 - d), crc_poly);
    }
  }));
  if (!d_mult) {
    return SRSRAN_ERROR;
  }
  turbo::WinParams p = {};
  p.input      = static_cast<const short*>(d_input);
  p.output     = d_output;
  p.ws         = h->d_ws;
  p.deint      = h->d_deint;
  p.inter      = h->d_inter;
  p.ws_stride  = h->ws_stride;
  p.K          = h->K;
  p.n_begin    = 0;
  p.n_end      = max_iterations;
  p.n_cb       = (int)n_cb;
  p.sb_layout  = sb_layout;
  p.in_is8     = in_is8 ? 1 : 0;
  p.desc       = d_desc;
  p.crc_poly   = crc_poly;
  p.crc_mult   = d_mult;
  p.noi        = d_noi;
  p.crc_ok     = d_crc_ok;
  if (want_lat(h, n_cb, 0)) {
    // a subframe's worth of code blocks: one block per wave instead of eight (turbo_lat_kernels.hip)
    if (h->arena) {
      h->d_ws_lat = static_cast<uint32_t*>(h->arena->ensure((size_t)turbo::lat_ws_dwords(h->K, h->nb) * ((n_cb + 1u) & ~1u) * sizeof(uint32_t), stream));
      if (!h->d_ws_lat) {
        return SRSRAN_ERROR;
      }
    } else if (ensure_lat_ws(h, n_cb)) {
      return SRSRAN_ERROR;
    }
    // the latency kernel forms the CRC from 32-bit words of its hard-bit image (sub-block d at d * sbs4 bytes): word (d, w) holds the steps
    // 32 w ... of sub-block d and is shifted into place with x^(bits behind it) mod g
    const uint32_t Wl = h->K / h->nb, nblkl = (Wl + 7) / 8, wps = (((nblkl + 1 + 3) & ~3u) >> 2);
    uint32_t*      d_ml = static_cast<uint32_t*>(cached_const(ckey(CK_CRC_LAT, Kq, h->nb, crc_poly), (size_t)nbq * wps * sizeof(uint32_t), [&](void* dst) {
      uint32_t* m = static_cast<uint32_t*>(dst);
      memset(m, 0, (size_t)nbq * wps * sizeof(uint32_t));
      for (uint32_t d = 0; d < nbq; d++) {
        for (uint32_t w = 0; w < wps; w++) {
          if (32 * w < Wl) {
            const uint32_t nbit = Wl - 32 * w > 32 ? 32 : Wl - 32 * w;
            m[d * wps + w]      = xpow_mod((uint64_t)Kq - ((uint64_t)d * Wl + 32 * w + nbit), crc_poly);
          }
        }
      }
    }));
    if (!d_ml) {
      return SRSRAN_ERROR;
    }
    p.crc_mult  = d_ml;
    p.ws        = h->d_ws_lat;
    p.ws_stride = turbo::lat_ws_dwords(h->K, h->nb);
    PHY_HIP_CHECK(want_lat2(h, n_cb) ? turbo::launch_lat2(h->arith8, p, stream) : turbo::launch_lat(h->nb, h->arith8, p, stream), SRSRAN_ERROR);
    h->state_in_lat = true;
    return SRSRAN_SUCCESS;
  }
  h->state_in_lat = false;
  if (h->arena) {
    const uint32_t cpw = 64 / (nbq / 2);
    p.ws = static_cast<uint32_t*>(h->arena->ensure((size_t)h->ws_stride * ceil_div(n_cb, cpw) * cpw * sizeof(uint32_t), stream));
    if (!p.ws) {
      return SRSRAN_ERROR;
    }
  }
  PHY_HIP_CHECK(turbo::launch_win(h->nb, h->arith8, p, stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

void* phyhip::turbo::WsArena::ensure(size_t bytes, hipStream_t stream)
{
  if (bytes <= cap) {
    return p;
  }
  if (p) {
    (void)hipStreamSynchronize(stream); // a launch enqueued earlier may still be working in the old arena
    (void)hipFree(p);
    p   = nullptr;
    cap = 0;
  }
  const size_t want = bytes + bytes / 4;
  if (hipMalloc(&p, want) != hipSuccess) {
    p = nullptr;
    set_error("turbo decoder: workspace allocation of %zu bytes failed", want);
    return nullptr;
  }
  cap = want;
  return p;
}
phyhip::turbo::WsArena::~WsArena()
{
  (void)hipFree(p);
}

int phyhip::turbo::batch_create_shared(srsran_hip_tdec_batch** h, uint32_t long_cb, uint32_t max_nof_cb, bool llr8_api, WsArena* arena)
{
  return tdec_batch_create(h, long_cb, max_nof_cb, SRSRAN_TDEC_AUTO, llr8_api, arena);
}

// all 188 block sizes x the decoders AUTO selects for 16- and 8-bit soft bits: exchange / interleaver tables and the CRC multipliers of both
// generators, collected in one host image and placed with one allocation and one upload
bool phyhip::turbo::prebuild_tables()
{
  struct Done {
    std::mutex mu;
    int        state = 0; // 0 not yet, 1 ok, -1 failed
  };
  Done&                       done = device_local<Done>();
  std::lock_guard<std::mutex> dl(done.mu);
  if (done.state) {
    return done.state > 0;
  }
  bool ok = false;
  {
    ConstCache& c = ccache();
    {
      std::lock_guard<std::mutex> lk(c.mu);
      c.bulk       = true;
      c.bulk_owner = std::this_thread::get_id();
    }
    const uint32_t polys[2] = {0x1864CFBu, 0x1800063u}; // CRC24A, CRC24B
    for (int i = 0; i < LTE_QPP_NOF_SIZES; i++) {
      const uint32_t K = lte_qpp_table[i][0];
      for (int api8 = 0; api8 < 2; api8++) {
        srsran_hip_tdec_batch_t* b = nullptr;
        WsArena                  none; // (no workspace is needed to collect tables)
        if (tdec_batch_create(&b, K, 1, SRSRAN_TDEC_AUTO, api8 != 0, &none) != SRSRAN_SUCCESS) {
          continue;
        }
        if (!b->nb) {
          for (uint32_t poly : polys) {
            (void)gen_crc_mult8(K, poly);
          }
        }
        if (b->nb) {
          const uint32_t nbq = (uint32_t)b->nb;
          for (uint32_t poly : polys) {
            (void)cached_const(ckey(CK_CRC_WIN, K, b->nb, poly), nbq * sizeof(uint32_t), [&](void* dst) {
              uint32_t* m = static_cast<uint32_t*>(dst);
              for (uint32_t d = 0; d < nbq; d++) {
                m[d] = xpow_mod((uint64_t)(K / nbq) * (uint64_t)(nbq - 1 - d), poly);
              }
            });
            if (turbo::lat_exists(b->nb, b->arith8)) {
              const uint32_t Wl = K / nbq, nblkl = (Wl + 7) / 8, wps = (((nblkl + 1 + 3) & ~3u) >> 2);
              (void)cached_const(ckey(CK_CRC_LAT, K, b->nb, poly), (size_t)nbq * wps * sizeof(uint32_t), [&](void* dst) {
                uint32_t* m = static_cast<uint32_t*>(dst);
                memset(m, 0, (size_t)nbq * wps * sizeof(uint32_t));
                for (uint32_t d = 0; d < nbq; d++) {
                  for (uint32_t w = 0; w < wps; w++) {
                    if (32 * w < Wl) {
                      const uint32_t nbit = Wl - 32 * w > 32 ? 32 : Wl - 32 * w;
                      m[d * wps + w]      = xpow_mod((uint64_t)K - ((uint64_t)d * Wl + 32 * w + nbit), poly);
                    }
                  }
                }
              });
            }
          }
        }
        delete b; // (its table pointers are placeholders: nothing to free)
      }
    }
    std::lock_guard<std::mutex> lk(c.mu);
    c.bulk = false;
    ok     = true;
    if (!c.pending.empty()) {
      void* d = nullptr;
      if (hipMalloc(&d, c.img.size()) != hipSuccess || upload(d, c.img.data(), c.img.size()) != hipSuccess) {
        (void)hipFree(d);
        ok = false;
      } else {
        for (auto& pd : c.pending) {
          c.dev[pd.first] = static_cast<uint8_t*>(d) + pd.second;
        }
      }
    }
    c.pending.clear();
    c.img.clear();
    c.img.shrink_to_fit();
  }
  done.state = ok ? 1 : -1;
  return ok;
}

// ------------------------------------------------------------------------------------------------ handle ABI

namespace {
struct TdecCtx {
  DeviceTag   tag;
  hipStream_t stream = nullptr;
  // one batch object (max_cb = 1) per (K, nb), created on first use
  std::map<uint64_t, srsran_hip_tdec_batch_t*> dec;
  int16_t* d_in  = nullptr; // 3*(Kmax+32)+12
  uint8_t* d_out = nullptr; // Kmax/8
  int16_t* h_in  = nullptr; // pinned
  uint8_t* h_out = nullptr; // pinned
  size_t   in_cap = 0;
  bool     dev_state_valid = false; // the handle's own batch object holds the decoder state of the current code block
};

TdecCtx* ctx_raw(srsran_tdec_t* h)
{
  return reinterpret_cast<TdecCtx*>(h->dec16_hdlr[0]);
}
// nullptr (error reported) when the decoder lives on another device than the calling thread's
TdecCtx* ctx_of(srsran_tdec_t* h)
{
  TdecCtx* c = ctx_raw(h);
  return (c && !check_device(c->tag, "srsran_tdec")) ? nullptr : c;
}
} // namespace

extern "C" int srsran_tdec_init(srsran_tdec_t* h, uint32_t max_long_cb)
{
  return srsran_tdec_init_manual(h, max_long_cb, SRSRAN_TDEC_AUTO);
}

extern "C" int srsran_tdec_init_manual(srsran_tdec_t* h, uint32_t max_long_cb, srsran_tdec_impl_type_t dec_type)
{
  memset(h, 0, sizeof(srsran_tdec_t)); // turbodecoder.c:150
  h->current_llr_type = SRSRAN_TDEC_16;
  switch (dec_type) {
    case SRSRAN_TDEC_AUTO:
    case SRSRAN_TDEC_GENERIC:
    case SRSRAN_TDEC_SSE_WINDOW:
    case SRSRAN_TDEC_AVX_WINDOW:
      break;
    case SRSRAN_TDEC_SSE8_WINDOW:
    case SRSRAN_TDEC_AVX8_WINDOW:
      h->current_llr_type = SRSRAN_TDEC_8; // turbodecoder.c:170-172,190-192
      break;
    default:
      // the non-window SSE decoder and the NEON one are not reproduced by the HIP engine
      fprintf(stderr, "Error decoder %d not supported\n", dec_type);
      return SRSRAN_ERROR;
  }
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  h->dec_type      = dec_type;
  h->max_long_cb   = max_long_cb;
  h->current_cbidx = -1;
  if (dec_type == SRSRAN_TDEC_AUTO) {
    // what the reference's tdec_init() of gen / sse16win / avx16win / sse8win / avx8win return (turbodecoder.c:252-270)
    h->nof_blocks16[0] = 1;
    h->nof_blocks16[1] = 8;
    h->nof_blocks16[2] = 16;
    h->nof_blocks8[0]  = 16;
    h->nof_blocks8[1]  = 32;
  } else if (dec_type == SRSRAN_TDEC_SSE8_WINDOW || dec_type == SRSRAN_TDEC_AVX8_WINDOW) {
    h->nof_blocks8[0] = dec_type == SRSRAN_TDEC_SSE8_WINDOW ? 16 : 32;
  } else {
    h->nof_blocks16[0] = dec_type == SRSRAN_TDEC_GENERIC ? 1 : (dec_type == SRSRAN_TDEC_SSE_WINDOW ? 8 : 16);
  }
  auto* c   = new TdecCtx;
  c->in_cap = 3 * ((size_t)max_long_cb + 32) + 12;
  PHY_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMalloc(&c->d_in, c->in_cap * sizeof(int16_t)), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMalloc(&c->d_out, max_long_cb / 8 + 8), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipHostMalloc(&c->h_in, c->in_cap * sizeof(int16_t)), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipHostMalloc(&c->h_out, max_long_cb / 8 + 8), SRSRAN_ERROR);
  h->dec16_hdlr[0] = c;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_tdec_free(srsran_tdec_t* h)
{
  TdecCtx* c = ctx_raw(h);
  if (c) {
    for (auto& kv : c->dec) {
      srsran_hip_tdec_batch_free(kv.second);
    }
    hipFree(c->d_in);
    hipFree(c->d_out);
    hipHostFree(c->h_in);
    hipHostFree(c->h_out);
    if (c->stream) {
      hipStreamDestroy(c->stream);
    }
    delete c;
  }
  memset(h, 0, sizeof(srsran_tdec_t)); // turbodecoder.c:362
}

extern "C" void srsran_tdec_force_not_sb(srsran_tdec_t* h)
{
  h->force_not_sb = true;
}

extern "C" int srsran_tdec_new_cb(srsran_tdec_t* h, uint32_t long_cb)
{
  if (long_cb > h->max_long_cb) {
    fprintf(stderr, "TDEC was initialized for max_long_cb=%d\n", h->max_long_cb);
    return -1;
  }
  if (TdecCtx* c = ctx_raw(h)) {
    c->dev_state_valid = false;
  }
  h->n_iter          = 0;
  h->current_long_cb = long_cb;
  h->current_cbidx   = srsran_cbsegm_cbindex(long_cb);
  if (h->current_cbidx < 0) {
    fprintf(stderr, "Invalid CB length %d\n", long_cb);
    return -1;
  }
  return 0;
}

extern "C" int srsran_tdec_get_nof_iterations(srsran_tdec_t* h)
{
  return h->n_iter;
}

// half iterations [n_iter, n_end) on the device + hard decision into `output` (turbodecoder.c:455-533).
// ELEM is the caller's LLR type: int16_t (srsran_tdec_iteration / run_all) or int8_t (the *_8bit entry points).
template <typename ELEM>
static void tdec_handle_iterate(srsran_tdec_t* h, ELEM* input, uint8_t* output, uint32_t n_end)
{
  constexpr bool in8 = sizeof(ELEM) == 1;
  TdecCtx*       c   = ctx_of(h);
  if (!c) {
    fprintf(stderr, "[srsran_phy_hip] srsran_tdec: handle not initialised\n");
    return;
  }
  bind_thread();
  const uint32_t K = h->current_long_cb;
  if ((uint32_t)srsran_cbsegm_cbsize(h->current_cbidx) != K) {
    fprintf(stderr, "[srsran_phy_hip] srsran_tdec: K=%u is not a valid turbo block size\n", K);
    return;
  }
  int  nb     = 0;
  bool arith8 = false;
  impl_to_cfg(h->dec_type, in8, K, &nb, &arith8);
  // Input layout (turbodecoder_iter.h:88): the 8-bit decoders always expect the rm_turbo sub-block layout, the
  // 16-bit ones only in AUTO mode with a window decoder; srsran_tdec_force_not_sb() turns it off.
  const bool auto_mode = h->dec_type == SRSRAN_TDEC_AUTO;
  const int  sb_layout = (!h->force_not_sb && (arith8 || (auto_mode && nb > 0))) ? 1 : 0;
  if (auto_mode) {
    h->current_llr_type = arith8 ? SRSRAN_TDEC_8 : SRSRAN_TDEC_16;
    h->current_dec      = arith8 ? (nb == 32 ? 1 : 0) : (nb == 16 ? 2 : (nb == 8 ? 1 : 0));
  } else {
    h->current_dec = 0;
  }
  h->current_inter_idx = nb == 32 ? 3 : (nb == 16 ? 2 : (nb == 8 ? 1 : 0)); // interleaver_idx(), turbodecoder.c:134-148

  uint64_t key = ((uint64_t)K << 8) | ((uint64_t)nb << 1) | (arith8 ? 1u : 0u);
  auto     it  = c->dec.find(key);
  if (it == c->dec.end()) {
    srsran_hip_tdec_batch_t* b = nullptr;
    int impl = arith8 ? (nb == 32 ? SRSRAN_TDEC_AVX8_WINDOW : SRSRAN_TDEC_SSE8_WINDOW)
                      : (nb == 16 ? SRSRAN_TDEC_AVX_WINDOW : (nb == 8 ? SRSRAN_TDEC_SSE_WINDOW : SRSRAN_TDEC_GENERIC));
    if (srsran_hip_tdec_batch_create(&b, K, 1, impl)) {
      fprintf(stderr, "[srsran_phy_hip] srsran_tdec: %s\n", get_error());
      return;
    }
    it = c->dec.emplace(key, b).first;
  }
  // a run that went through the shared submission queue left no decoder state in this handle's private object: a caller that
  // resumes it with srsran_tdec_iteration gets the earlier half iterations re-run from its input first
  const uint32_t n_begin = c->dev_state_valid ? (uint32_t)h->n_iter : 0u;
  const size_t   in_len  = sb_layout ? 3 * ((size_t)K + 32) + 12 : 3 * (size_t)K + 12;
  if (n_begin == 0) {
    // The reference converts between LLR widths on the host when API and decoder differ (convert_8_to_16 /
    // convert_16_to_8, turbodecoder.c:443-453); here the kernel's extraction does it.  (The reference
    // converts only 3K+12 elements even for the longer sub-block layout; all of it is converted here.)
    memcpy(c->h_in, input, in_len * sizeof(ELEM));
    if (sb_layout && in8 == arith8) {
      // the reference writes the tail into the caller's buffer here (turbodecoder_iter.h:58-70,92-96)
      for (uint32_t i = K; i < K + 3; i++) {
        input[i]                = input[3 * (K + 32) + 2 * (i - K)];
        input[K + 32 + i]       = input[3 * (K + 32) + 2 * (i - K) + 1];
        input[2 * (K + 32) + i] = input[3 * (K + 32) + 6 + 2 * (i - K) + 1];
      }
    }
    PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->d_in, c->h_in, in_len * sizeof(ELEM), hipMemcpyHostToDevice, c->stream));
  }
  if (tdec_batch_run_range(it->second, c->d_in, in8, (uint32_t)in_len, c->d_out, K / 8, 1, n_begin, n_end, sb_layout, false,
                           c->stream)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_tdec: %s\n", get_error());
    return;
  }
  PHY_HIP_CHECK_VOID(hipMemcpyAsync(c->h_out, c->d_out, K / 8, hipMemcpyDeviceToHost, c->stream));
  PHY_HIP_CHECK_VOID(hipStreamSynchronize(c->stream));
  memcpy(output, c->h_out, K / 8);
  h->n_iter          = (int)n_end;
  c->dev_state_valid = true;
}

// srsran_tdec_run_all{,_8bit}: a whole run from the input is stateless, so runs of the same shape that are in flight on
// different handles (one worker thread per subframe, cc_worker.cc:212-231) share one batch launch (coalesce.h)
template <typename ELEM>
static bool tdec_run_all_queued(srsran_tdec_t* h, ELEM* input, uint8_t* output, uint32_t nit)
{
  constexpr bool in8 = sizeof(ELEM) == 1;
  TdecCtx*       c   = ctx_of(h);
  const uint32_t K   = h->current_long_cb;
  if (!c || !coalescing_enabled() || (uint32_t)srsran_cbsegm_cbsize(h->current_cbidx) != K) {
    return false;
  }
  // handles are often initialised on one thread and run on another: the queue's lanes (streams, staging buffers, engines) must be
  // created on the process's device even when this worker thread has not touched the device yet
  bind_thread();
  int  nb     = 0;
  bool arith8 = false;
  if (impl_to_cfg(h->dec_type, in8, K, &nb, &arith8) || (nb && (K % nb || K / nb <= 40))) {
    return false; // the private path reports the error
  }
  const bool   auto_mode = h->dec_type == SRSRAN_TDEC_AUTO;
  const int    sb_layout = (!h->force_not_sb && (arith8 || (auto_mode && nb > 0))) ? 1 : 0;
  const size_t in_len    = sb_layout ? 3 * ((size_t)K + 32) + 12 : 3 * (size_t)K + 12;
  char         key[96];
  snprintf(key, sizeof(key), "tdec:d%d:K%u:nb%d:a%d:sb%d:e%d:it%u", current_device(), K, nb, arith8 ? 1 : 0, sb_layout, in8 ? 1 : 0, nit);
  std::shared_ptr<Coalescer> q = coalescer_for(key, [&]() -> Coalescer* {
    const uint32_t cap  = 64;
    const int      impl = arith8 ? (nb == 32 ? SRSRAN_TDEC_AVX8_WINDOW : SRSRAN_TDEC_SSE8_WINDOW)
                                 : (nb == 16 ? SRSRAN_TDEC_AVX_WINDOW : (nb == 8 ? SRSRAN_TDEC_SSE_WINDOW : SRSRAN_TDEC_GENERIC));
    const uint32_t in_stride  = (uint32_t)(Coalescer::stride_of(in_len * sizeof(ELEM)) / sizeof(ELEM));
    const uint32_t out_stride = (uint32_t)Coalescer::stride_of(K / 8);
    return new Coalescer(in_len * sizeof(ELEM), K / 8, cap, 4, [=](int) -> Coalescer::Engine {
      srsran_hip_tdec_batch_t* b = nullptr;
      if (srsran_hip_tdec_batch_create(&b, K, cap, impl)) {
        return Coalescer::Engine();
      }
      return Coalescer::Engine{[=](const void* d_in, void* d_out, uint32_t n, uint64_t, hipStream_t st) {
                                 return tdec_batch_run_range(b, d_in, in8, in_stride, static_cast<uint8_t*>(d_out), out_stride, n, 0, nit, sb_layout,
                                                             false, st);
                               },
                               [=]() { srsran_hip_tdec_batch_free(b); }};
    });
  });
  if (!q) {
    return false;
  }
  if (auto_mode) {
    h->current_llr_type = arith8 ? SRSRAN_TDEC_8 : SRSRAN_TDEC_16;
    h->current_dec      = arith8 ? (nb == 32 ? 1 : 0) : (nb == 16 ? 2 : (nb == 8 ? 1 : 0));
  } else {
    h->current_dec = 0;
  }
  h->current_inter_idx = nb == 32 ? 3 : (nb == 16 ? 2 : (nb == 8 ? 1 : 0));
  if (q->submit(input, output) != SRSRAN_SUCCESS) {
    fprintf(stderr, "[srsran_phy_hip] srsran_tdec_run_all: %s\n", get_error());
    return false;
  }
  if (sb_layout && in8 == arith8) {
    // the reference writes the tail into the caller's buffer (turbodecoder_iter.h:58-70,92-96)
    for (uint32_t i = K; i < K + 3; i++) {
      input[i]                = input[3 * (K + 32) + 2 * (i - K)];
      input[K + 32 + i]       = input[3 * (K + 32) + 2 * (i - K) + 1];
      input[2 * (K + 32) + i] = input[3 * (K + 32) + 6 + 2 * (i - K) + 1];
    }
  }
  h->n_iter          = (int)nit;
  c->dev_state_valid = false;
  return true;
}

extern "C" void srsran_tdec_iteration(srsran_tdec_t* h, int16_t* input, uint8_t* output)
{
  if (h->current_cbidx >= 0) {
    tdec_handle_iterate(h, input, output, (uint32_t)h->n_iter + 1);
  }
}

extern "C" int srsran_tdec_run_all(srsran_tdec_t* h, int16_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb)
{
  if (srsran_tdec_new_cb(h, long_cb)) {
    return SRSRAN_ERROR;
  }
  if (tdec_run_all_queued(h, input, output, nof_iterations ? nof_iterations : 1)) {
    return SRSRAN_SUCCESS;
  }
  tdec_handle_iterate(h, input, output, nof_iterations ? nof_iterations : 1);
  return h->n_iter ? SRSRAN_SUCCESS : SRSRAN_ERROR;
}

extern "C" void srsran_tdec_iteration_8bit(srsran_tdec_t* h, int8_t* input, uint8_t* output)
{
  if (h->current_cbidx >= 0) {
    tdec_handle_iterate(h, input, output, (uint32_t)h->n_iter + 1);
  }
}

extern "C" int srsran_tdec_run_all_8bit(srsran_tdec_t* h, int8_t* input, uint8_t* output, uint32_t nof_iterations, uint32_t long_cb)
{
  if (srsran_tdec_new_cb(h, long_cb)) {
    return SRSRAN_ERROR;
  }
  if (tdec_run_all_queued(h, input, output, nof_iterations ? nof_iterations : 1)) {
    return SRSRAN_SUCCESS;
  }
  tdec_handle_iterate(h, input, output, nof_iterations ? nof_iterations : 1);
  return h->n_iter ? SRSRAN_SUCCESS : SRSRAN_ERROR;
}
