// rm_host.cpp -- host side of the turbo rate de-matching: position tables, device table pool, drop-in + batch API.
//
// Mirrors (interface + behaviour) lib/src/phy/fec/turbo/rm_turbo.c of the reference.  The position tables are
// derived here from TS 36.212 5.1.4.1 directly (sub-block interleavers, bit collection into the circular buffer,
// selection from k0 skipping the <NULL> entries) rather than from the reference's table builders; the oracle pins
// both against the compiled reference.
#include "hip_common.h"
#include "rm_device.h"
#include "srsran_amd/phy_sch_abi.h"

#include <map>
#include <mutex>
#include <vector>

using namespace phyhip;

namespace {

const uint8_t kColPerm[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
                              1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31}; // 36.212 table 5.1.4-1

// positions, in the receiver's buffer, of the 3(K+4) soft bits of one redundancy version in transmission order
//   natural buffer (nof_sb == 0): stream s in {0,1,2}, index i in [0, K+4)  ->  3 i + s
//   decoder layout (nof_sb > 0) : i < K: s (K+32) + (i mod W) nof_sb + i / W with W = K / nof_sb; tail: 3 (K+32) + ...
std::vector<uint16_t> build_table(uint32_t K, uint32_t rv, uint32_t nof_sb, uint32_t* start_index = nullptr)
{
  const int D = (int)K + 4, R = (D + 31) / 32, Kp = 32 * R, ND = Kp - D;
  // circular buffer w: stream / index of every entry, -1 = <NULL>
  std::vector<int> w(3 * Kp, -1);
  for (int k = 0; k < Kp; k++) {
    const int col = kColPerm[k / R], row = k % R;
    const int y01 = col + 32 * row;            // sub-block interleaver of d0 and d1
    const int y2  = (col + 32 * row + 1) % Kp; // and of d2
    if (y01 >= ND) {
      w[k]          = 3 * (y01 - ND);
      w[Kp + 2 * k] = 3 * (y01 - ND) + 1;
    }
    if (y2 >= ND) {
      w[Kp + 2 * k + 1] = 3 * (y2 - ND) + 2;
    }
  }
  const int Ncb = 3 * Kp;
  const int k0  = R * (2 * ((Ncb + 8 * R - 1) / (8 * R)) * (int)rv + 2);
  if (start_index) { // place of the first transmitted bit in the buffer without its <NULL> entries (k0_vec[..][1] in rm_turbo.c)
    uint32_t r0 = 0;
    for (int j = 0; j < k0; j++) {
      r0 += w[j] >= 0;
    }
    *start_index = r0 % (uint32_t)(3 * D);
  }
  std::vector<uint16_t> t;
  t.reserve(3 * D);
  for (int j = 0; (int)t.size() < 3 * D; j++) {
    const int v = w[(k0 + j) % Ncb];
    if (v < 0) {
      continue;
    }
    if (!nof_sb) {
      t.push_back((uint16_t)v);
    } else if ((uint32_t)v < 3 * K) {
      const uint32_t s = (uint32_t)v % 3, i = (uint32_t)v / 3, W = K / nof_sb;
      t.push_back((uint16_t)(s * (K + 32) + (i % W) * nof_sb + i / W));
    } else {
      t.push_back((uint16_t)((uint32_t)v - 3 * K + 3 * (K + 32)));
    }
  }
  return t;
}

// device pool of tables, built on first use
struct TablePool {
  std::mutex                       mu;
  std::map<uint32_t, uint16_t*>    dev;  // key = K | rv << 16 | nof_sb << 20 (bit 31: forward table of the transmit side)
  std::map<uint32_t, uint32_t>     fwd_len;
  bool                             all = false; // build_all_tables() has run
  // (the tables are the process's: never freed -- entries may point into one big allocation, and the runtime may be gone at static destruction)
};
TablePool& g_pool_ref()
{
  return device_local<TablePool>(); // the tables of the calling thread's device
}
#define g_pool (g_pool_ref())

const uint16_t* table_on_device(uint32_t K, uint32_t rv, uint32_t nof_sb)
{
  const uint32_t              key = K | (rv << 16) | (nof_sb << 20);
  std::lock_guard<std::mutex> lk(g_pool.mu);
  auto                        it = g_pool.dev.find(key);
  if (it != g_pool.dev.end()) {
    return it->second;
  }
  // the device holds the INVERSE table (soft-buffer position -> index of the transmitted bit, 0xffff = none): the
  // kernel is driven from the output side (unit-stride read-modify-write of the soft buffer)
  const std::vector<uint16_t> fwd = build_table(K, rv, nof_sb);
  std::vector<uint16_t>       t(nof_sb ? 3 * ((size_t)K + 32) + 12 : 3 * (size_t)K + 12, 0xffffu);
  for (size_t k = 0; k < fwd.size(); k++) {
    t[fwd[k]] = (uint16_t)k;
  }
  uint16_t* d = nullptr;
  if (hipMalloc(&d, t.size() * sizeof(uint16_t)) != hipSuccess ||
      upload(d, t.data(), t.size() * sizeof(uint16_t)) != hipSuccess) {
    set_error("rm_turbo: cannot place the table of K=%u rv=%u on the device", K, rv);
    (void)hipFree(d);
    return nullptr;
  }
  g_pool.dev[key] = d;
  return d;
}

// forward table of the natural buffer (transmit side): bit k of the rate-matched output = d[table[k mod len]]
const uint16_t* fwd_table_on_device(uint32_t K, uint32_t rv, uint32_t* len)
{
  const uint32_t              key = K | (rv << 16) | (1u << 31);
  std::lock_guard<std::mutex> lk(g_pool.mu);
  auto                        it = g_pool.dev.find(key);
  if (it != g_pool.dev.end()) {
    *len = g_pool.fwd_len[key];
    return it->second;
  }
  const std::vector<uint16_t> fwd = build_table(K, rv, 0);
  uint16_t*                   d   = nullptr;
  if (hipMalloc(&d, fwd.size() * sizeof(uint16_t)) != hipSuccess ||
      upload(d, fwd.data(), fwd.size() * sizeof(uint16_t)) != hipSuccess) {
    set_error("rm_turbo: cannot place the transmit table of K=%u rv=%u on the device", K, rv);
    (void)hipFree(d);
    return nullptr;
  }
  g_pool.dev[key]     = d;
  g_pool.fwd_len[key] = (uint32_t)fwd.size();
  *len                = (uint32_t)fwd.size();
  return d;
}

int rx_batch(const void* d_in, uint32_t in_stride, uint32_t in_len, void* d_out, uint32_t out_stride, uint32_t n_cb, uint32_t K,
             uint32_t rv, uint32_t nof_sb, bool elem8, hipStream_t st)
{
  const int idx = srsran_cbsegm_cbindex(K);
  if (n_cb == 0 && rv <= 3 && idx >= 0) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!d_in || !d_out || rv > 3 || idx < 0 || (uint32_t)srsran_cbsegm_cbsize(idx) != K ||
      (nof_sb && (K % nof_sb || (nof_sb != 8 && nof_sb != 16 && nof_sb != 32)))) {
    set_error("rm_turbo batch: invalid arguments (K=%u rv=%u nof_sb=%u)", K, rv, nof_sb);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  const uint16_t* tab = table_on_device(K, rv, nof_sb);
  if (!tab) {
    return SRSRAN_ERROR;
  }
  const rm::RxJob first = {0, in_len, 0, 3 * K + 12, 0};
  const uint32_t  span  = nof_sb ? 3 * (K + 32) + 12 : 3 * K + 12;
  PHY_HIP_CHECK(rm::launch_rx_gather(d_in, d_out, tab, span, nullptr, first, in_stride, out_stride, (int)n_cb, elem8, st, in_len), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

// host-pointer form: one code block.  Staging buffers and the stream are kept per calling thread (the reference's function
// is stateless; allocating per call would dominate it).
struct HostStage {
  hipStream_t st   = nullptr;
  void*       d_in = nullptr;
  void*       d_out = nullptr;
  size_t      cap_in = 0, cap_out = 0;
  bool        tried = false;
  ~HostStage()
  {
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (st) {
      (void)hipStreamDestroy(st);
    }
  }
  bool ready()
  {
    if (!tried) {
      tried = true;
      if (device_available() && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        st = nullptr;
      }
    }
    return st != nullptr;
  }
  static bool grow(void** p, size_t* cap, size_t need)
  {
    if (need <= *cap) {
      return true;
    }
    (void)hipFree(*p);
    *p   = nullptr;
    *cap = 0;
    if (hipMalloc(p, need + 256) != hipSuccess) {
      return false;
    }
    *cap = need + 256;
    return true;
  }
};

template <typename T>
int rx_host(const T* input, T* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv, uint32_t nof_sb)
{
  if (rv >= 4 || cb_idx >= 188) {
    printf("Invalid inputs rv_idx=%d, cb_idx=%d\n", rv, cb_idx);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  HostStage& s = thread_device_local<HostStage>();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_rm_turbo_rx_lut: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  if (in_len == 0) {
    return SRSRAN_SUCCESS;
  }
  const uint32_t K = (uint32_t)srsran_cbsegm_cbsize(cb_idx);
  const size_t   n_out = nof_sb ? 3 * ((size_t)K + 32) + 12 : 3 * (size_t)K + 12;
  if (!HostStage::grow(&s.d_in, &s.cap_in, in_len * sizeof(T)) || !HostStage::grow(&s.d_out, &s.cap_out, n_out * sizeof(T))) {
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(s.d_in, input, in_len * sizeof(T), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(s.d_out, output, n_out * sizeof(T), hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  const int rc = rx_batch(s.d_in, in_len, in_len, s.d_out, (uint32_t)n_out, 1, K, rv, nof_sb, sizeof(T) == 1, s.st);
  if (rc != SRSRAN_SUCCESS) {
    fprintf(stderr, "[srsran_phy_hip] srsran_rm_turbo_rx_lut: %s\n", get_error());
    return rc;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(output, s.d_out, n_out * sizeof(T), hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

} // namespace

// rm_turbo.c:276-340 builds every table of every block size here, and srsran_sch_init calls it (sch.c:166): the init-time hook of the transport-block
// path.  All 188 sizes x 4 redundancy versions x {16-bit decoder layout, 8-bit decoder layout, transmit side} in one allocation and one upload
// (about 40 MB, 0.1 s), then one worker's staging contexts warmed (chan_host.cpp: srsran_hip_warmup).
extern "C" int srsran_hip_warmup(uint32_t nof_workers);
extern "C" void srsran_rm_turbo_gentables(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) == hipSuccess && n > 0) { // (without a device every later call fails loudly by itself)
    (void)srsran_hip_warmup(1); // idempotent per device
  }
}

// srsran_sch_free calls this for EVERY object (sch.c:201) while other objects may still be decoding: the tables stay (they are the process's)
extern "C" void srsran_rm_turbo_free_tables(void) {}

namespace phyhip {
namespace rm {
bool build_all_tables()
{
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (g_pool.all) {
    return true;
  }
  struct Item {
    uint32_t key;
    size_t   off, len;
  };
  std::vector<Item>     items;
  std::vector<uint16_t> img;
  auto                  add = [&](uint32_t key, const std::vector<uint16_t>& t) {
    if (g_pool.dev.count(key)) {
      return;
    }
    const size_t off = (img.size() + 63) & ~(size_t)63; // 128-byte aligned tables
    img.resize(off);
    img.insert(img.end(), t.begin(), t.end());
    items.push_back({key, off, t.size()});
  };
  for (int i = 0; i < 188; i++) {
    const uint32_t K      = (uint32_t)srsran_cbsegm_cbsize((uint32_t)i);
    const uint32_t nsb[2] = {srsran_tdec_autoimp_get_subblocks(K), srsran_tdec_autoimp_get_subblocks_8bit(K)};
    for (uint32_t rv = 0; rv < 4; rv++) {
      for (int v = 0; v < 2; v++) {
        if (v == 1 && nsb[1] == nsb[0]) {
          continue;
        }
        const std::vector<uint16_t> fwd = build_table(K, rv, nsb[v]);
        std::vector<uint16_t>       t(nsb[v] ? 3 * ((size_t)K + 32) + 12 : 3 * (size_t)K + 12, 0xffffu);
        for (size_t k = 0; k < fwd.size(); k++) {
          t[fwd[k]] = (uint16_t)k;
        }
        add(K | (rv << 16) | (nsb[v] << 20), t);
      }
      const std::vector<uint16_t> f = build_table(K, rv, 0);
      add(K | (rv << 16) | (1u << 31), f);
      g_pool.fwd_len[K | (rv << 16) | (1u << 31)] = (uint32_t)f.size();
    }
  }
  if (!items.empty()) {
    uint16_t* d = nullptr;
    if (hipMalloc(&d, img.size() * sizeof(uint16_t)) != hipSuccess || upload(d, img.data(), img.size() * sizeof(uint16_t)) != hipSuccess) {
      set_error("rm_turbo: cannot place the rate-matching tables on the device");
      (void)hipFree(d);
      return false;
    }
    for (const Item& it : items) {
      g_pool.dev[it.key] = d + it.off;
    }
  }
  g_pool.all = true;
  return true;
}
} // namespace rm
} // namespace phyhip

extern "C" int srsran_rm_turbo_rx_lut_(int16_t* input, int16_t* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv_idx,
                                       bool enable_input_tdec)
{
  // rm_turbo.c:412-421: the decoder layout is used when the 16-bit AUTO decoder for this K is a window decoder
  const uint32_t nsb = (enable_input_tdec && cb_idx < 188) ? srsran_tdec_autoimp_get_subblocks((uint32_t)srsran_cbsegm_cbsize(cb_idx)) : 0;
  return rx_host(input, output, in_len, cb_idx, rv_idx, nsb);
}

extern "C" int srsran_rm_turbo_rx_lut(int16_t* input, int16_t* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv_idx)
{
  return srsran_rm_turbo_rx_lut_(input, output, in_len, cb_idx, rv_idx, true);
}

extern "C" int srsran_rm_turbo_rx_lut_8bit(int8_t* input, int8_t* output, uint32_t in_len, uint32_t cb_idx, uint32_t rv_idx)
{
  const uint32_t nsb = cb_idx < 188 ? srsran_tdec_autoimp_get_subblocks_8bit((uint32_t)srsran_cbsegm_cbsize(cb_idx)) : 0;
  return rx_host(input, output, in_len, cb_idx, rv_idx, nsb);
}

extern "C" int srsran_hip_rm_turbo_rx_batch(const int16_t* d_input, uint32_t in_stride, uint32_t in_len, int16_t* d_softbuf,
                                            uint32_t out_stride, uint32_t n_cb, uint32_t long_cb, uint32_t rv_idx,
                                            uint32_t nof_sb, void* stream)
{
  return rx_batch(d_input, in_stride, in_len, d_softbuf, out_stride, n_cb, long_cb, rv_idx, nof_sb, false, (hipStream_t)stream);
}

extern "C" int srsran_hip_rm_turbo_rx_batch_8bit(const int8_t* d_input, uint32_t in_stride, uint32_t in_len, int8_t* d_softbuf,
                                                 uint32_t out_stride, uint32_t n_cb, uint32_t long_cb, uint32_t rv_idx,
                                                 uint32_t nof_sb, void* stream)
{
  return rx_batch(d_input, in_stride, in_len, d_softbuf, out_stride, n_cb, long_cb, rv_idx, nof_sb, true, (hipStream_t)stream);
}

// the position table itself (host memory, 3K+12 entries): what the reference keeps in its static deinterleaver /
// deinterleaver_sb arrays (rm_turbo.c:77-101)
extern "C" int srsran_hip_rm_turbo_table(uint16_t* table, uint32_t long_cb, uint32_t rv_idx, uint32_t nof_sb)
{
  const int idx = srsran_cbsegm_cbindex(long_cb);
  if (!table || rv_idx > 3 || idx < 0 || (uint32_t)srsran_cbsegm_cbsize(idx) != long_cb ||
      (nof_sb && (long_cb % nof_sb || (nof_sb != 8 && nof_sb != 16 && nof_sb != 32)))) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  std::vector<uint16_t> t = build_table(long_cb, rv_idx, nof_sb);
  memcpy(table, t.data(), t.size() * sizeof(uint16_t));
  return SRSRAN_SUCCESS;
}

// used by the transport-block decoder (sch_host.cpp): device table of (K, rv, layout)
namespace phyhip {
namespace rm {
const uint16_t* device_table(uint32_t K, uint32_t rv, uint32_t nof_sb)
{
  return table_on_device(K, rv, nof_sb);
}
const uint16_t* device_fwd_table(uint32_t K, uint32_t rv, uint32_t* len)
{
  return fwd_table_on_device(K, rv, len);
}
uint32_t tx_start_index(uint32_t K, uint32_t rv)
{
  uint32_t r0 = 0;
  (void)build_table(K, rv, 0, &r0);
  return r0;
}
std::vector<uint16_t> host_table(uint32_t K, uint32_t rv, uint32_t nof_sb)
{
  return build_table(K, rv, nof_sb);
}
} // namespace rm
} // namespace phyhip
