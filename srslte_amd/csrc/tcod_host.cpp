// tcod_host.cpp -- C ABI of the LTE turbo encoder and of the transmit side of a transport block
// (include/srsran_amd/phy_sch_abi.h): srsran_tcod_{init,free,encode}, srsran_hip_tcod_encode_batch, srsran_hip_sch_encode.
#include "hip_common.h"
#include "srsran_amd/phy_sch_abi.h"
#include "tables/lte_qpp_table.h"
#include "tcod_device.h"
#include "sch_stage.h"

#include <algorithm>
#include <map>
#include <vector>

using namespace phyhip;

namespace phyhip {
namespace rm {
const uint16_t* device_fwd_table(uint32_t K, uint32_t rv, uint32_t* len); // rm_host.cpp
uint32_t        tx_start_index(uint32_t K, uint32_t rv);
}
} // namespace phyhip

namespace {

bool qpp_params(uint32_t K, uint32_t* f1, uint32_t* f2)
{
  const int idx = srsran_cbsegm_cbindex(K);
  if (idx < 0 || (uint32_t)srsran_cbsegm_cbsize((uint32_t)idx) != K) {
    return false;
  }
  *f1 = lte_qpp_table[idx][1];
  *f2 = lte_qpp_table[idx][2];
  return true;
}

struct Ctx { // behind srsran_tcod_t.temp
  DeviceTag tag;
  hipStream_t st    = nullptr;
  uint8_t*    d_in  = nullptr;
  uint8_t*    d_out = nullptr;
};

} // namespace

extern "C" int srsran_hip_tcod_encode_batch(const uint8_t* d_in, uint32_t in_stride, uint8_t* d_out, uint32_t out_stride, uint32_t n_cb,
                                            uint32_t long_cb, void* stream)
{
  tcod::EncParams p{};
  if (!d_in || !d_out || !qpp_params(long_cb, &p.f1, &p.f2) || in_stride < long_cb || out_stride < 3 * long_cb + 12) {
    set_error("tcod batch: invalid arguments (K=%u)", long_cb);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  p.in         = d_in;
  p.out        = d_out;
  p.in_stride  = in_stride;
  p.out_stride = out_stride;
  p.n_cb       = n_cb;
  p.K          = long_cb;
  PHY_HIP_CHECK(tcod::launch_encode(p, (hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

// turbocoder.c:40-75
extern "C" int srsran_tcod_init(srsran_tcod_t* h, uint32_t max_long_cb)
{
  if (!h) {
    return -1;
  }
  h->max_long_cb = max_long_cb;
  h->temp        = nullptr;
  if (!device_available()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_tcod_init: %s (there is no CPU fallback)\n", get_error());
    return -1;
  }
  Ctx* c = new Ctx;
  if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->d_in, 6144 + 64) != hipSuccess ||
      hipMalloc(&c->d_out, 3 * 6144 + 64) != hipSuccess) {
    (void)hipFree(c->d_in);
    (void)hipFree(c->d_out);
    delete c;
    return -1;
  }
  h->temp = reinterpret_cast<uint8_t*>(c);
  return 0;
}

extern "C" void srsran_tcod_free(srsran_tcod_t* h)
{
  if (!h) {
    return;
  }
  h->max_long_cb = 0;
  if (h->temp) {
    Ctx* c = reinterpret_cast<Ctx*>(h->temp);
    (void)hipFree(c->d_in);
    (void)hipFree(c->d_out);
    (void)hipStreamDestroy(c->st);
    delete c;
    h->temp = nullptr;
  }
}

extern "C" int srsran_tcod_encode(srsran_tcod_t* h, uint8_t* input, uint8_t* output, uint32_t long_cb)
{
  if (long_cb > h->max_long_cb) {
    fprintf(stderr, "Turbo coder initiated for max_long_cb=%d\n", h->max_long_cb); // turbocoder.c:85-88
    return -1;
  }
  uint32_t f1, f2;
  if (!qpp_params(long_cb, &f1, &f2)) {
    fprintf(stderr, "Invalid CB size %d\n", long_cb); // turbocoder.c:90-94
    return -1;
  }
  Ctx* c = reinterpret_cast<Ctx*>(h->temp);
  if (!c) {
    return -1;
  }
  PHY_DEV_GUARD(c->tag, "srsran_tcod_encode", -1);
  PHY_HIP_CHECK(hipMemcpyAsync(c->d_in, input, long_cb, hipMemcpyHostToDevice, c->st), -1);
  if (srsran_hip_tcod_encode_batch(c->d_in, long_cb, c->d_out, 3 * long_cb + 12, 1, long_cb, c->st) != SRSRAN_SUCCESS) {
    return -1;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(output, c->d_out, 3 * long_cb + 12, hipMemcpyDeviceToHost, c->st), -1);
  PHY_HIP_CHECK(hipStreamSynchronize(c->st), -1);
  return 0;
}

// ------------------------------------------------------------------------------------------------ byte-packed per-block entry points
extern "C" void srsran_tcod_gentable(void) {} // turbocoder.c:345-400: nothing to precompute here

// turbocoder.c:188-343.  crc_tb carries the running transport-block checksum from block to block (only its crcinit is read and
// written, masked to the CRC's order); crc_cb, when given, is left holding the code-block checksum.
extern "C" int srsran_tcod_encode_lut(srsran_tcod_t* h, srsran_crc_t* crc_tb, srsran_crc_t* crc_cb, uint8_t* input, uint8_t* parity,
                                      uint32_t cblen_idx, bool last_cb)
{
  if (cblen_idx >= 188 || !h || !crc_tb || !input || !parity) {
    return -1;
  }
  const uint32_t K = (uint32_t)srsran_cbsegm_cbsize(cblen_idx);
  Ctx*           c = reinterpret_cast<Ctx*>(h->temp);
  if (!c) {
    return -1;
  }
  PHY_DEV_GUARD(c->tag, "srsran_tcod_encode_lut", -1);
  if (crc_tb->order != 24 || (crc_cb && crc_cb->order != 24)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_tcod_encode_lut: only 24-bit CRCs are supported\n");
    return -1;
  }
  tcod::LutEncParams p{};
  if (!qpp_params(K, &p.f1, &p.f2)) {
    return -1;
  }
  const uint32_t crc_bits = (crc_cb ? 24u : 0u) + (last_cb ? 24u : 0u);
  if (K < crc_bits) {
    return -1;
  }
  p.K            = K;
  p.n_data_bytes = (K - crc_bits) / 8;
  p.in           = c->d_in;
  p.out_sys      = c->d_out;
  p.out_par      = c->d_out + 1024;
  p.crc_state    = reinterpret_cast<uint32_t*>(c->d_out + 4096);
  p.tb_poly      = (uint32_t)crc_tb->polynom;
  p.cb_poly      = crc_cb ? (uint32_t)crc_cb->polynom : 0;
  p.has_cb_crc   = crc_cb ? 1 : 0;
  p.last_cb      = last_cb ? 1 : 0;
  uint32_t state[2] = {(uint32_t)(crc_tb->crcinit & crc_tb->crcmask), 0};
  PHY_HIP_CHECK(hipMemcpyAsync(c->d_in, input, p.n_data_bytes, hipMemcpyHostToDevice, c->st), -1);
  PHY_HIP_CHECK(hipMemcpyAsync(p.crc_state, state, sizeof(state), hipMemcpyHostToDevice, c->st), -1);
  PHY_HIP_CHECK(tcod::launch_lut_encode(p, c->st), -1);
  PHY_HIP_CHECK(hipMemcpyAsync(input, p.out_sys, K / 8 + 1, hipMemcpyDeviceToHost, c->st), -1);
  PHY_HIP_CHECK(hipMemcpyAsync(parity, p.out_par, K / 4 + 1, hipMemcpyDeviceToHost, c->st), -1);
  PHY_HIP_CHECK(hipMemcpyAsync(state, p.crc_state, sizeof(state), hipMemcpyDeviceToHost, c->st), -1);
  PHY_HIP_CHECK(hipStreamSynchronize(c->st), -1);
  crc_tb->crcinit = state[0];
  if (crc_cb) {
    crc_cb->crcinit = state[1];
  }
  return (int)(3 * K + 12);
}

namespace {
struct TxStage {
  hipStream_t st   = nullptr;
  uint8_t*    d_sp = nullptr; // systematic + parity streams
  uint8_t*    d_out = nullptr;
  size_t      cap_out = 0;
  bool        tried = false;
  ~TxStage()
  {
    (void)hipFree(d_sp);
    (void)hipFree(d_out);
    if (st) {
      (void)hipStreamDestroy(st);
    }
  }
  bool ready()
  {
    if (!tried) {
      tried = true;
      if (device_available() && (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&d_sp, 4096) != hipSuccess)) {
        st = nullptr;
      }
    }
    return st != nullptr;
  }
};

// srsran_bit_copy (bit.c:685-698) of `len` bits from the packed array `src` (bit 0 first) to dst at bit offset `dst_off`.  With
// both offsets byte aligned in the reference (src_aligned tells whether its source offset was) whole bytes are copied and the
// unused low bits of a last partial byte are ZEROED; otherwise the bits outside the range are preserved (bitarray_copy).
void bit_copy_like_reference(uint8_t* dst, uint32_t dst_off, const uint8_t* src, uint32_t src_off, uint32_t len, bool src_aligned)
{
  auto get = [&](uint32_t i) { return (src[(src_off + i) >> 3] >> (7u - ((src_off + i) & 7u))) & 1u; };
  for (uint32_t i = 0; i < len; i++) {
    const uint32_t o = dst_off + i;
    dst[o >> 3]      = (uint8_t)((dst[o >> 3] & ~(0x80u >> (o & 7u))) | (get(i) << (7u - (o & 7u))));
  }
  if ((dst_off & 7u) == 0 && src_aligned && (len & 7u)) {
    dst[(dst_off + len) >> 3] &= (uint8_t)(0xff00u >> (len & 7u));
  }
}
} // namespace

// rm_turbo.c:340-378.  w_buff keeps the block's byte-packed streams between redundancy versions (its layout is private to
// this function, as the reference's is to its own): [systematic K/8 + 1 bytes | parity K/4 + 1 bytes].
extern "C" int srsran_rm_turbo_tx_lut(uint8_t* w_buff, uint8_t* systematic, uint8_t* parity, uint8_t* output, uint32_t cb_idx, uint32_t out_len,
                                      uint32_t w_offset, uint32_t rv_idx)
{
  if (rv_idx >= 4 || cb_idx >= 188 || !w_buff || !output) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  TxStage& s = thread_device_local<TxStage>();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] srsran_rm_turbo_tx_lut: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  const uint32_t K = (uint32_t)srsran_cbsegm_cbsize(cb_idx), in_len = 3 * K + 12;
  const uint32_t nsys = K / 8 + 1, npar = K / 4 + 1;
  if (rv_idx == 0) {
    if (!systematic || !parity) {
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    memcpy(w_buff, systematic, nsys);
    memcpy(w_buff + nsys, parity, npar);
  }
  if (out_len == 0) {
    return SRSRAN_SUCCESS;
  }
  const size_t out_bytes = (out_len + 7) / 8;
  if (out_bytes > s.cap_out) {
    (void)hipFree(s.d_out);
    s.d_out   = nullptr;
    s.cap_out = 0;
    PHY_HIP_CHECK(hipMalloc(&s.d_out, out_bytes + 1024), SRSRAN_ERROR);
    s.cap_out = out_bytes + 1024;
  }
  tcod::LutRmParams p{};
  p.table = rm::device_fwd_table(K, rv_idx, &p.table_len);
  if (!p.table) {
    return SRSRAN_ERROR;
  }
  p.sys = s.d_sp;
  p.par = s.d_sp + nsys;
  p.out = s.d_out;
  p.K   = K;
  p.E   = out_len;
  std::vector<uint8_t> bits(out_bytes);
  PHY_HIP_CHECK(hipMemcpyAsync(s.d_sp, w_buff, nsys + npar, hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(tcod::launch_lut_rm(p, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMemcpyAsync(bits.data(), s.d_out, out_bytes, hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  // the reference copies the circular buffer piece by piece (rm_turbo.c:362-374); only the alignment rule of each piece matters here
  uint32_t w_len = 0, r_ptr = rm::tx_start_index(K, rv_idx);
  while (w_len < out_len) {
    uint32_t cp = out_len - w_len;
    if (cp + r_ptr >= in_len) {
      cp = in_len - r_ptr;
    }
    bit_copy_like_reference(output, w_len + w_offset, bits.data(), w_len, cp, (r_ptr & 7u) == 0);
    r_ptr += cp;
    if (r_ptr >= in_len) {
      r_ptr -= in_len;
    }
    w_len += cp;
  }
  return SRSRAN_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ transport blocks, transmit side
struct srsran_hip_sch_enc {
  DeviceTag tag;
  void*  d_scratch = nullptr;
  void*  h_scratch = nullptr; // pinned
  size_t cap       = 0;
  hipEvent_t done  = nullptr;
  bool   pending   = false;
  // lane multipliers of the two CRCs, one row of 64 per message length seen so far (key: length | generator << 31), resident on the device
  std::map<uint32_t, uint32_t> mult_row;
  uint32_t*                    d_mult = nullptr;
  uint32_t*                    h_mult = nullptr; // pinned mirror: a new row goes up by an asynchronous copy on the call's stream, in front of the kernel
  uint32_t                     mult_rows_cap = 0;
  // the latency kernel's rows of 256 lane multipliers (tcod_device.h: crc_lane_multipliers256), same scheme
  std::map<uint32_t, uint32_t> mult256_row;
  uint32_t*                    d_mult256 = nullptr;
  uint32_t*                    h_mult256 = nullptr;
  uint32_t                     mult256_rows_cap = 0;
};

namespace {
uint32_t h_mulmod24(uint32_t a, uint32_t m, uint32_t poly)
{
  uint32_t r = 0;
  for (int i = 23; i >= 0; i--) {
    r = ((r << 1) & 0xffffffu) ^ (((r >> 23) & 1u) ? poly : 0u);
    r ^= ((m >> i) & 1u) ? a : 0u;
  }
  return r;
}
uint32_t h_xpow24(uint32_t n, uint32_t poly)
{
  uint32_t r = 1, b = 2;
  while (n) {
    if (n & 1u) {
      r = h_mulmod24(r, b, poly);
    }
    b = h_mulmod24(b, b, poly);
    n >>= 1;
  }
  return r;
}
} // namespace

void phyhip::tcod::crc_lane_multipliers(uint32_t n_units, uint32_t bits_per_unit, uint32_t poly24, uint32_t* m64)
{
  const uint32_t L = (n_units + 63u) / 64u;
  for (uint32_t l = 0; l < 64; l++) {
    const uint32_t i1 = std::min((l + 1) * L, n_units);
    m64[l]            = h_xpow24(bits_per_unit * (n_units - i1), poly24);
  }
}

void phyhip::tcod::crc_lane_multipliers256(uint32_t n_units, uint32_t bits_per_unit, uint32_t poly24, uint32_t* m256)
{
  // m[l] = (x^(bits per stretch))^(255 - l): one exponentiation, then one multiplication per lane
  const uint32_t L    = (n_units + 255u) / 256u;
  const uint32_t step = h_xpow24(bits_per_unit * L, poly24);
  m256[255]           = 1;
  for (int l = 254; l >= 0; l--) {
    m256[l] = h_mulmod24(m256[l + 1], step, poly24);
  }
}

// row of lane multipliers for a message of n units (bytes for the transport CRC24A, bits for the code-block CRC24B): computed and uploaded
// the first time the length is seen by this object
static uint32_t enc_mult_row(srsran_hip_sch_enc_t* h, uint32_t n_units, bool cb, hipStream_t st)
{
  const uint32_t key = n_units | (cb ? 0x80000000u : 0u);
  auto           it  = h->mult_row.find(key);
  if (it == h->mult_row.end()) {
    const uint32_t row = (uint32_t)h->mult_row.size();
    if (row >= h->mult_rows_cap) {
      const uint32_t cap = h->mult_rows_cap ? 2 * h->mult_rows_cap : 32;
      uint32_t *     nd = nullptr, *nh = nullptr;
      if (hipMalloc(&nd, (size_t)cap * 64 * sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&nh, (size_t)cap * 64 * sizeof(uint32_t)) != hipSuccess ||
          hipDeviceSynchronize() != hipSuccess || // nothing may still read the old table / copy from the old mirror when they are freed below
          (h->d_mult && (hipMemcpy(nd, h->d_mult, (size_t)row * 64 * sizeof(uint32_t), hipMemcpyDeviceToDevice) != hipSuccess ||
                         hipDeviceSynchronize() != hipSuccess))) {
        (void)hipFree(nd);
        (void)hipHostFree(nh);
        set_error("sch encode: device allocation of the CRC multiplier table failed");
        return 0xffffffffu;
      }
      if (h->h_mult) {
        memcpy(nh, h->h_mult, (size_t)row * 64 * sizeof(uint32_t));
      }
      (void)hipFree(h->d_mult);
      (void)hipHostFree(h->h_mult);
      h->d_mult        = nd;
      h->h_mult        = nh;
      h->mult_rows_cap = cap;
    }
    uint32_t* m = h->h_mult + (size_t)row * 64;
    tcod::crc_lane_multipliers(n_units, cb ? 1u : 8u, cb ? 0x800063u : 0x864CFBu, m);
    // (stream order puts the row in front of the kernel that reads it; the pinned source stays where it is)
    if (hipMemcpyAsync(h->d_mult + (size_t)row * 64, m, 64 * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess) {
      set_error("sch encode: upload of the CRC multiplier table failed");
      return 0xffffffffu;
    }
    it = h->mult_row.emplace(key, row).first;
  }
  return it->second; // (a row INDEX: the table may move while a call's jobs are still being written)
}

static uint32_t enc_mult256_row(srsran_hip_sch_enc_t* h, uint32_t n_units, bool cb, hipStream_t st)
{
  const uint32_t key = n_units | (cb ? 0x80000000u : 0u);
  auto           it  = h->mult256_row.find(key);
  if (it == h->mult256_row.end()) {
    const uint32_t row = (uint32_t)h->mult256_row.size();
    if (row >= h->mult256_rows_cap) {
      const uint32_t cap = h->mult256_rows_cap ? 2 * h->mult256_rows_cap : 16;
      uint32_t *     nd = nullptr, *nh = nullptr;
      if (hipMalloc(&nd, (size_t)cap * 256 * sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&nh, (size_t)cap * 256 * sizeof(uint32_t)) != hipSuccess ||
          hipDeviceSynchronize() != hipSuccess ||
          (h->d_mult256 && (hipMemcpy(nd, h->d_mult256, (size_t)row * 256 * sizeof(uint32_t), hipMemcpyDeviceToDevice) != hipSuccess ||
                            hipDeviceSynchronize() != hipSuccess))) {
        (void)hipFree(nd);
        (void)hipHostFree(nh);
        set_error("sch encode: device allocation of the CRC multiplier table failed");
        return 0xffffffffu;
      }
      if (h->h_mult256) {
        memcpy(nh, h->h_mult256, (size_t)row * 256 * sizeof(uint32_t));
      }
      (void)hipFree(h->d_mult256);
      (void)hipHostFree(h->h_mult256);
      h->d_mult256        = nd;
      h->h_mult256        = nh;
      h->mult256_rows_cap = cap;
    }
    uint32_t* m = h->h_mult256 + (size_t)row * 256;
    tcod::crc_lane_multipliers256(n_units, cb ? 1u : 8u, cb ? 0x800063u : 0x864CFBu, m);
    if (hipMemcpyAsync(h->d_mult256 + (size_t)row * 256, m, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess) {
      set_error("sch encode: upload of the CRC multiplier table failed");
      return 0xffffffffu;
    }
    it = h->mult256_row.emplace(key, row).first;
  }
  return it->second;
}

extern "C" int srsran_hip_sch_enc_create(srsran_hip_sch_enc_t** hh)
{
  if (!hh) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  srsran_hip_sch_enc* h = new srsran_hip_sch_enc;
  if (hipEventCreateWithFlags(&h->done, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return SRSRAN_ERROR;
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_sch_enc_free(srsran_hip_sch_enc_t* h)
{
  if (!h) {
    return;
  }
  if (h->pending) {
    (void)hipEventSynchronize(h->done);
  }
  (void)hipFree(h->d_scratch);
  (void)hipHostFree(h->h_scratch);
  (void)hipFree(h->d_mult);
  (void)hipHostFree(h->h_mult);
  (void)hipFree(h->d_mult256);
  (void)hipHostFree(h->h_mult256);
  (void)hipEventDestroy(h->done);
  delete h;
}

extern "C" int srsran_hip_sch_encode(srsran_hip_sch_enc_t* h, const uint8_t* d_data, const srsran_hip_tb_t* tbs, uint32_t n_tb, uint8_t* d_e_bits,
                                     void* stream)
{
  TraceRange trace_("srsran_hip_sch_encode");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_sch_encode", SRSRAN_ERROR);
  }
  if (h && n_tb == 0) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!h || !d_data || !tbs || !d_e_bits) {
    set_error("sch encode: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t st = (hipStream_t)stream;
  // Two passes with nothing allocated per code block (32 k of them per call in the benches): pass 1 segments and validates the transport
  // blocks and counts the code blocks, pass 2 writes the jobs straight into the pinned image of the device array.  Interleaver parameters and
  // rate-matching tables are looked up once per (block size, rv) of a transport block, not once per code block.
  std::vector<srsran_cbsegm_t> seg(n_tb);
  size_t                       n_cb = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_hip_tb_t& tb = tbs[t];
    if (srsran_cbsegm(&seg[t], tb.tbs) || tb.Qm == 0 || tb.rv > 3 || (tb.tbs & 7) || tb.tbs == 0 || tb.nof_e_bits % tb.Qm) {
      set_error("sch encode: transport block %u: invalid tbs / Qm / rv / nof_e_bits", t);
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    if (seg[t].F) {
      fprintf(stderr, "Error filler bits are not supported. Use standard TBS\n"); // sch.c:249-252
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    n_cb += seg[t].C;
  }
  const size_t bytes = n_cb * sizeof(tcod::TbCbJob) + n_tb * (sizeof(tcod::TbCrcJob) + 4) + 64;
  if (h->pending) {
    PHY_HIP_CHECK(hipEventSynchronize(h->done), SRSRAN_ERROR);
    h->pending = false;
  }
  if (bytes > h->cap) {
    (void)hipFree(h->d_scratch);
    (void)hipHostFree(h->h_scratch);
    h->d_scratch = h->h_scratch = nullptr;
    h->cap                      = 0;
    PHY_HIP_CHECK(hipMalloc(&h->d_scratch, bytes * 2), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipHostMalloc(&h->h_scratch, bytes * 2), SRSRAN_ERROR);
    h->cap = bytes * 2;
  }
  uint8_t* hb   = static_cast<uint8_t*>(h->h_scratch);
  auto*    cbs  = reinterpret_cast<tcod::TbCbJob*>(hb);
  auto*    crcs = reinterpret_cast<tcod::TbCrcJob*>(hb + n_cb * sizeof(tcod::TbCbJob));
  struct KInfo { // what every code block of one size and rv shares; the last look-up is kept (a batch is mostly one block size)
    uint32_t        K = 0, rv = 0xffffffffu, f1 = 0, f2 = 0, table_len = 0;
    const uint16_t* table = nullptr;
  };
  KInfo cache[2];
  int  err  = SRSRAN_SUCCESS;
  auto info = [&](uint32_t K, uint32_t rv, int slot) -> const KInfo* {
    KInfo& c = cache[slot];
    if (c.K != K || c.rv != rv) {
      c.K  = K;
      c.rv = 0xffffffffu; // (not valid until both look-ups have succeeded)
      if (!qpp_params(K, &c.f1, &c.f2)) {
        err = SRSRAN_ERROR_INVALID_INPUTS;
        return nullptr;
      }
      c.table = rm::device_fwd_table(K, rv, &c.table_len);
      if (!c.table) {
        err = SRSRAN_ERROR;
        return nullptr;
      }
      c.rv = rv;
    }
    return &c;
  };
  // A subframe's worth of code blocks or less: ONE launch, a workgroup of 256 lanes per code block (tcod_kernels.hip: tb_encode_lat_kernel) -- the
  // throughput kernels give a block to one wave and need a launch of their own for the transport CRCs (SRSRAN_HIP_TCOD_LAT=0 keeps them)
  const bool lat = n_cb <= 64 && knob(KNOB_TCOD_LAT) != 0;
  size_t at = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_hip_tb_t& tb = tbs[t];
    const srsran_cbsegm_t& cs = seg[t];
    crcs[t]                   = {tb.data_offset, tb.tbs / 8, lat ? 0u : enc_mult_row(h, tb.tbs / 8, false, st), lat ? enc_mult256_row(h, tb.tbs / 8, false, st) : 0u};
    if (crcs[t].crc_mult_row == 0xffffffffu || crcs[t].crc_mult256_row == 0xffffffffu) {
      return SRSRAN_ERROR;
    }
    // sch.c:254-330: bits per block and rate-matched lengths
    const uint32_t Gp = tb.nof_e_bits / tb.Qm, gamma = Gp % cs.C;
    uint32_t       src = 8 * tb.data_offset, wp = tb.e_offset;
    for (uint32_t i = 0; i < cs.C; i++) {
      // sch.c:284-290: the transmit side puts the C2 smaller blocks first (the receive side, sch.c:392, the C1 larger ones; valid LTE block sizes never mix the two)
      const bool   small = i < cs.C2;
      const KInfo* ki    = info(small ? cs.K2 : cs.K1, tb.rv, small ? 1 : 0);
      if (!ki) {
        return err;
      }
      tcod::TbCbJob  j{};
      j.K                 = ki->K;
      const uint32_t rlen = cs.C == 1 ? j.K : j.K - 24; // bits of the block without its own CRC
      const bool     last = i + 1 == cs.C;
      j.src_bit           = src;
      j.n_src_bits        = last ? rlen - 24 : rlen; // the last block ends with the 24 transport-block CRC bits
      j.tb_crc            = last ? t : 0xffffffffu;
      j.crc24b            = cs.C > 1 ? 1u : 0u;
      j.E                 = tb.Qm * (Gp / cs.C) + ((i <= cs.C - gamma - 1) ? 0u : tb.Qm); // sch.c:296-300
      j.out_bit           = wp;
      j.f1                = ki->f1;
      j.f2                = ki->f2;
      j.table             = ki->table;
      j.table_len         = ki->table_len;
      j.crc_mult_row      = (j.crc24b && !lat) ? enc_mult_row(h, j.n_src_bits + (last ? 24u : 0u), true, st) : 0u;
      j.crc_mult256_row   = (j.crc24b && lat) ? enc_mult256_row(h, j.n_src_bits + (last ? 24u : 0u), true, st) : 0u;
      if (j.crc_mult_row == 0xffffffffu || j.crc_mult256_row == 0xffffffffu) {
        return SRSRAN_ERROR;
      }
      src += j.n_src_bits;
      wp += j.E;
      cbs[at++] = j;
    }
  }
  uint8_t* db = static_cast<uint8_t*>(h->d_scratch);
  PHY_HIP_CHECK(hipMemcpyAsync(db, hb, n_cb * sizeof(tcod::TbCbJob) + n_tb * sizeof(tcod::TbCrcJob), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
  tcod::TbParams p{};
  p.data   = d_data;
  p.e_bits = d_e_bits;
  p.cbs    = reinterpret_cast<const tcod::TbCbJob*>(db);
  p.tbs    = reinterpret_cast<const tcod::TbCrcJob*>(db + n_cb * sizeof(tcod::TbCbJob));
  p.tb_crc = reinterpret_cast<uint32_t*>(db + n_cb * sizeof(tcod::TbCbJob) + n_tb * sizeof(tcod::TbCrcJob));
  p.n_cb   = (uint32_t)n_cb;
  p.n_tb   = n_tb;
  p.crc_mult = h->d_mult;
  p.crc_mult256 = h->d_mult256;
  // the code blocks OR their partial bytes into the output: clear every transport block's range first (adjacent ranges merged)
  std::vector<std::pair<uint32_t, uint32_t>> rng;
  for (uint32_t t = 0; t < n_tb; t++) {
    rng.emplace_back(tbs[t].e_offset / 8, (tbs[t].e_offset + tbs[t].nof_e_bits + 7) / 8);
  }
  std::sort(rng.begin(), rng.end());
  for (size_t i = 0; i < rng.size();) {
    uint32_t b0 = rng[i].first, b1 = rng[i].second;
    for (i++; i < rng.size() && rng[i].first <= b1; i++) {
      b1 = std::max(b1, rng[i].second);
    }
    PHY_HIP_CHECK(hipMemsetAsync(d_e_bits + b0, 0, b1 - b0, st), SRSRAN_ERROR);
  }
  if (lat) {
    PHY_HIP_CHECK(tcod::launch_tb_encode_lat(p, st), SRSRAN_ERROR);
  } else {
    PHY_HIP_CHECK(tcod::launch_tb_crc24a(p, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(tcod::launch_tb_encode(p, st), SRSRAN_ERROR);
  }
  PHY_HIP_CHECK(hipEventRecord(h->done, st), SRSRAN_ERROR);
  h->pending = true;
  return SRSRAN_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ the reference's transmit-side entry
//
// encode_tb (sch.c:239-368) as srsran_dlsch_encode2 (:625-658) reaches it, for ONE transport block on the caller's HOST buffers: payload bytes
// in, byte-packed rate-matched bits out.  The reference keeps every code block's coded bits in softbuffer->buffer_b[i] so that a call with
// data == NULL (a retransmission) only rate-matches again; here the rows keep the block's PAYLOAD slice instead (their contents are private
// to the encoder either way) and such a call encodes again from them.
namespace {
struct TxTbStage {
  hipStream_t           st  = nullptr;
  srsran_hip_sch_enc_t* enc = nullptr;
  uint8_t*              pin = nullptr; // pinned, device-visible image: [payload | e bits]
  uint8_t*              dev = nullptr; // e bits on the device (the code blocks OR their partial bytes into them: not a job for host memory)
  size_t                cap = 0;
  bool                  tried = false;
  ~TxTbStage()
  {
    srsran_hip_sch_enc_free(enc);
    (void)hipFree(dev);
    (void)hipHostFree(pin);
    if (st) {
      (void)hipStreamDestroy(st);
    }
  }
  bool ready()
  {
    if (!tried) {
      tried = true;
      if (device_available()) {
        bind_thread();
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
          st = nullptr;
        } else if (srsran_hip_sch_enc_create(&enc) != SRSRAN_SUCCESS) {
          (void)hipStreamDestroy(st);
          st = nullptr;
        }
      }
    }
    return st != nullptr;
  }
  bool grow(size_t need)
  {
    if (need <= cap) {
      return true;
    }
    (void)hipFree(dev);
    (void)hipHostFree(pin);
    dev = pin = nullptr;
    cap = 0;
    if (hipMalloc((void**)&dev, need) != hipSuccess || host_image_alloc(&pin, need) != hipSuccess) {
      return false;
    }
    cap = need;
    return true;
  }
};
} // namespace

extern "C" int srsran_hip_encode_tb(srsran_softbuffer_tx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits,
                                    uint8_t* data, uint8_t* e_bits)
{
  return phyhip::sch::encode_tb_staged(softbuffer, cb_segm, Qm, rv, nof_e_bits, data, e_bits, nullptr);
}

// n transport blocks, their e bits consumed on the device (sch_stage.h)
int phyhip::sch::encode_tbs_staged(TxItem* it, uint32_t n, const GroupBackEnd* back)
{
  if (!it || !back || n == 0) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  auto   al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t pay_bytes = 0, e_bytes = 0, n_cb = 0;
  for (uint32_t t = 0; t < n; t++) {
    TxItem& x = it[t];
    if (!x.seg || !x.sb) {
      fprintf(stderr, "Invalid parameters: e_bits=%d, cb_segm=%d, softbuffer=%d\n", 1, x.seg != 0, x.sb != 0); // sch.c:351
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    if (x.seg->F) {
      fprintf(stderr, "Error filler bits are not supported. Use standard TBS\n"); // :254-257
      return SRSRAN_ERROR;
    }
    if (x.seg->C > x.sb->max_cb) {
      fprintf(stderr, "Error number of CB to encode (%d) exceeds soft buffer size (%d CBs)\n", x.seg->C, x.sb->max_cb); // :259-262
      return SRSRAN_ERROR;
    }
    if (x.Qm == 0 || x.rv > 3 || x.seg->C == 0 || x.seg->tbs == 0 || x.nof_e_bits == 0) {
      fprintf(stderr, "Invalid Qm\n"); // :264-267
      return SRSRAN_ERROR;
    }
    pay_bytes += al(x.seg->tbs / 8 + 8);
    e_bytes += al((x.nof_e_bits + 7) / 8 + 8);
    n_cb += x.seg->C;
  }
  static thread_local StageRef<TxTbStage> ref;
  TxTbStage&                             s = ref.get();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] encode_tb: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  if (!s.grow(e_bytes + pay_bytes + 512)) { // pinned: [payloads]; device: [e bits | payloads]
    fprintf(stderr, "[srsran_phy_hip] encode_tb: staging allocation failed\n");
    return SRSRAN_ERROR;
  }
  std::vector<srsran_hip_tb_t> tbs(n);
  std::vector<uint32_t>        e_off(n);
  size_t                       po = 0, eo = 0;
  for (uint32_t t = 0; t < n; t++) {
    TxItem&        x = it[t];
    const uint32_t C = x.seg->C;
    uint32_t       rp = 0;
    for (uint32_t i = 0; i < C; i++) { // payload slices of the code blocks in the transmit side's order (the C2 smaller blocks first, sch.c:284-290)
      const uint32_t K    = i < x.seg->C2 ? x.seg->K2 : x.seg->K1;
      const uint32_t rlen = C > 1 ? K - 24 : K;
      const uint32_t nb   = (i + 1 == C ? rlen - 24 : rlen) / 8;
      if (!x.sb->buffer_b[i]) {
        return SRSRAN_ERROR;
      }
      if (x.data) {
        memcpy(x.sb->buffer_b[i], x.data + rp / 8, nb);
      }
      memcpy(s.pin + po + rp / 8, x.sb->buffer_b[i], nb);
      rp += 8 * nb;
    }
    if (rp != x.seg->tbs) {
      fprintf(stderr, "[srsran_phy_hip] encode_tb: segmentation does not add up to the transport block size (%u != %u)\n", rp, x.seg->tbs);
      return SRSRAN_ERROR;
    }
    tbs[t]       = {x.seg->tbs, x.Qm, x.rv, x.nof_e_bits, (uint32_t)(8 * eo), (uint32_t)po, 0};
    e_off[t]     = (uint32_t)eo;
    x.e_byte_off = (uint32_t)eo;
    po += al(x.seg->tbs / 8 + 8);
    eo += al((x.nof_e_bits + 7) / 8 + 8);
  }
  const bool direct = n_cb <= 64 && knob(KNOB_TCOD_LAT) != 0; // (as encode_tb_staged: the one-launch kernel reads the payload from the pinned image)
  if (!direct) {
    PHY_HIP_CHECK(hipMemcpyAsync(s.dev + e_bytes, s.pin, pay_bytes, hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  }
  if (srsran_hip_sch_encode(s.enc, direct ? s.pin : s.dev + e_bytes, tbs.data(), n, s.dev, s.st) != SRSRAN_SUCCESS) {
    (void)hipStreamSynchronize(s.st);
    fprintf(stderr, "[srsran_phy_hip] encode_tb: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  const bool ok = (*back)(s.st, s.dev, e_off.data(), n);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR); // (also after a failed enqueue: nothing may be in flight when the images are re-used)
  if (!ok) {
    fprintf(stderr, "[srsran_phy_hip] encode_tb: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  return SRSRAN_SUCCESS;
}

// `back` given: the e bits stay on the device and the kernels back(stream, d_e_bits) enqueues consume them there (chan_host.cpp: scrambling +
// modulation, or the UL channel interleaver) -- their results are the caller's to collect after this function's one host wait
int phyhip::sch::encode_tb_staged(srsran_softbuffer_tx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, uint8_t* data,
                                  uint8_t* e_bits, const BackEnd* back)
{
  if ((!e_bits && !back) || !cb_segm || !softbuffer) {
    fprintf(stderr, "Invalid parameters: e_bits=%d, cb_segm=%d, softbuffer=%d\n", e_bits != 0, cb_segm != 0, softbuffer != 0); // sch.c:351
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (cb_segm->F) {
    fprintf(stderr, "Error filler bits are not supported. Use standard TBS\n"); // :254-257
    return SRSRAN_ERROR;
  }
  if (cb_segm->C > softbuffer->max_cb) {
    fprintf(stderr, "Error number of CB to encode (%d) exceeds soft buffer size (%d CBs)\n", cb_segm->C, softbuffer->max_cb); // :259-262
    return SRSRAN_ERROR;
  }
  if (Qm == 0 || rv > 3) {
    fprintf(stderr, "Invalid Qm\n"); // :264-267
    return SRSRAN_ERROR;
  }
  const uint32_t C = cb_segm->C, tbs = cb_segm->tbs;
  if (C == 0 || tbs == 0 || nof_e_bits == 0) {
    return SRSRAN_SUCCESS; // the loop over code blocks does not run
  }
  static thread_local StageRef<TxTbStage> ref;
  TxTbStage&                             s = ref.get();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] encode_tb: %s (there is no CPU fallback)\n", get_error());
    return SRSRAN_ERROR;
  }
  auto         al    = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t n_out = (nof_e_bits + 7) / 8;
  const size_t o_pay = 0, o_e = al(o_pay + tbs / 8 + 8);
  const size_t d_pay = al(n_out + 8); // device image: [e bits | payload]
  if (!s.grow(al(o_e + n_out + 8) + 512)) {
    fprintf(stderr, "[srsran_phy_hip] encode_tb: staging allocation failed\n");
    return SRSRAN_ERROR;
  }
  // payload slices of the code blocks in the transmit side's order (the C2 smaller blocks first, sch.c:284-290)
  uint32_t rp = 0;
  for (uint32_t i = 0; i < C; i++) {
    const uint32_t K    = i < cb_segm->C2 ? cb_segm->K2 : cb_segm->K1;
    const uint32_t rlen = C > 1 ? K - 24 : K;
    const uint32_t nb   = (i + 1 == C ? rlen - 24 : rlen) / 8; // the last block ends with the transport-block CRC, which is not payload
    if (!softbuffer->buffer_b[i]) {
      return SRSRAN_ERROR;
    }
    if (data) {
      memcpy(softbuffer->buffer_b[i], data + rp / 8, nb);
    }
    memcpy(s.pin + o_pay + rp / 8, softbuffer->buffer_b[i], nb);
    rp += 8 * nb;
  }
  if (rp != tbs) {
    fprintf(stderr, "[srsran_phy_hip] encode_tb: segmentation does not add up to the transport block size (%u != %u)\n", rp, tbs);
    return SRSRAN_ERROR;
  }
  const srsran_hip_tb_t tb = {tbs, Qm, rv, nof_e_bits, 0, 0, 0};
  // The one-launch kernel of a transport block reads the payload ONCE, a dword per lane: straight from the pinned image.  (The throughput kernels read it
  // byte-wise and more than once, which is slow across the bus: for them -- more than 64 code blocks never happen here, SRSRAN_HIP_TCOD_LAT=0 does -- it goes up with a copy.)
  const bool direct = C <= 64 && knob(KNOB_TCOD_LAT) != 0;
  if (!direct) {
    PHY_HIP_CHECK(hipMemcpyAsync(s.dev + d_pay, s.pin + o_pay, tbs / 8, hipMemcpyHostToDevice, s.st), SRSRAN_ERROR);
  }
  if (srsran_hip_sch_encode(s.enc, direct ? s.pin + o_pay : s.dev + d_pay, &tb, 1, s.dev, s.st) != SRSRAN_SUCCESS) {
    (void)hipStreamSynchronize(s.st);
    fprintf(stderr, "[srsran_phy_hip] encode_tb: %s\n", get_error());
    return SRSRAN_ERROR;
  }
  if (back) {
    const bool ok = (*back)(s.st, s.dev);
    PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR); // (also after a failed enqueue: nothing may be in flight when the images are re-used)
    if (!ok) {
      fprintf(stderr, "[srsran_phy_hip] encode_tb: %s\n", get_error());
      return SRSRAN_ERROR;
    }
    return SRSRAN_SUCCESS;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(s.pin + o_e, s.dev, n_out, hipMemcpyDeviceToHost, s.st), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize(s.st), SRSRAN_ERROR);
  // The unused low bits of the last byte: srsran_rm_turbo_tx_lut copies every block's circular buffer piece by piece with srsran_bit_copy
  // (rm_turbo.c:362-374), which ZEROES the rest of a piece's last byte when the piece starts byte aligned on both sides and preserves it
  // otherwise (bit.c:685-698): replay the pieces of the last code block for the byte the transport block ends in.
  const uint32_t tail = nof_e_bits & 7u;
  uint8_t        keep = 0;
  if (tail) {
    const uint32_t Gp = nof_e_bits / Qm, gamma = Gp % C;
    const uint32_t K  = C - 1 < cb_segm->C2 ? cb_segm->K2 : cb_segm->K1;
    const uint32_t E  = Qm * (Gp / C) + ((C - 1 <= C - gamma - 1) ? 0u : Qm);
    const uint32_t wp = nof_e_bits - E, in_len = 3 * K + 12;
    bool           zeroed = false;
    uint32_t       w_len = 0, r_ptr = rm::tx_start_index(K, rv);
    while (w_len < E) {
      uint32_t cp = E - w_len;
      if (cp + r_ptr >= in_len) {
        cp = in_len - r_ptr;
      }
      const uint32_t d0 = wp + w_len;
      if (((d0 + cp) >> 3) == (nof_e_bits >> 3)) {
        if ((d0 & 7u) == 0 && (r_ptr & 7u) == 0 && (cp & 7u)) {
          zeroed = true;
        }
      }
      r_ptr += cp;
      if (r_ptr >= in_len) {
        r_ptr -= in_len;
      }
      w_len += cp;
    }
    keep = zeroed ? 0 : (uint8_t)(e_bits[n_out - 1] & (0xffu >> tail));
  }
  memcpy(e_bits, s.pin + o_e, n_out);
  if (tail) {
    e_bits[n_out - 1] = (uint8_t)((e_bits[n_out - 1] & (0xff00u >> tail)) | keep);
  }
  return SRSRAN_SUCCESS;
}
