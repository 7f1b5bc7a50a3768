// sch_host.cpp -- transport-block decoding on the device: srsran_cbsegm, and decode_tb / decode_tb_cb of
// lib/src/phy/phch/sch.c:370-560 for a batch of transport blocks (rate de-matching -> turbo half iterations with
// CRC early stop per code block -> transport-block CRC).
#include "hip_common.h"
#include "rm_device.h"
#include "srsran_amd/phy_sch_abi.h"
#include "turbo_device.h"
#include "sch_stage.h"

#include <algorithm>
#include <map>
#include <vector>

using namespace phyhip;

namespace phyhip {
namespace rm {
const uint16_t* device_table(uint32_t K, uint32_t rv, uint32_t nof_sb); // inverse table, see rm_host.cpp
}
} // namespace phyhip

#define CRC24A 0x1864CFBu // crc.h:41-42
#define CRC24B 0x1800063u

// ------------------------------------------------------------------------------------------------ segmentation

extern "C" int srsran_cbsegm(srsran_cbsegm_t* s, uint32_t tbs)
{
  memset(s, 0, sizeof(*s));
  if (tbs == 0) {
    return SRSRAN_SUCCESS;
  }
  // 36.212 5.1.2 with Z = 6144: B = tbs + 24 (transport-block CRC); more than one block -> 24 more bits each
  const uint32_t B = tbs + 24, Z = 6144;
  uint32_t       C = 1, Bp = B;
  if (B > Z) {
    C  = (B + (Z - 24) - 1) / (Z - 24);
    Bp = B + 24 * C;
  }
  const int i1 = srsran_cbsegm_cbindex((Bp - 1) / C + 1); // smallest K with C K >= B'
  if (i1 < 0) {
    return SRSRAN_ERROR;
  }
  s->tbs    = tbs;
  s->C      = C;
  s->K1     = (uint32_t)srsran_cbsegm_cbsize((uint32_t)i1);
  s->K1_idx = (uint32_t)i1;
  if (C == 1) {
    s->C1 = 1;
  } else {
    // cbsegm.c:85-101: the next smaller size (the same size when K1 is the smallest one)
    s->K2_idx = i1 > 0 ? (uint32_t)i1 - 1 : 0;
    s->K2     = (uint32_t)srsran_cbsegm_cbsize(i1 > 0 ? (uint32_t)i1 - 1 : (uint32_t)i1);
    s->C2     = s->K1 != s->K2 ? (C * s->K1 - Bp) / (s->K1 - s->K2) : 0;
    s->C1     = C - s->C2;
  }
  s->L_tb = 24;
  s->L_cb = 24;
  s->F    = s->C1 * s->K1 + s->C2 * s->K2 - Bp;
  return SRSRAN_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ batch decoder

struct srsran_hip_sch {
  DeviceTag tag;
  // one turbo batch object per (K, arithmetic is 16 bit), grown on demand
  std::map<uint32_t, std::pair<srsran_hip_tdec_batch_t*, uint32_t>> dec; // K | 8-bit flag << 31 -> (object, capacity)
  turbo::WsArena arena;       // the one workspace all its decoders run in (turbo_device.h)
  void*  d_scratch = nullptr; // job / descriptor / result arrays
  void*  h_scratch = nullptr; // ... and their pinned host image: descriptors go up and verdicts come down with one asynchronous copy each
  size_t scratch_cap = 0;
  // transport-block CRC: one row of 256 lane multipliers per block size seen so far (rm::tb_crc_multipliers), resident on the device
  std::map<uint32_t, uint32_t> crc_row; // tbs -> row
  uint32_t*                    d_crc_mult = nullptr;
  uint32_t*                    h_crc_mult = nullptr; // pinned mirror: a new row goes up asynchronously on the call's stream, in front of the CRC kernel
  uint32_t                     crc_rows_cap = 0;
};

extern "C" int srsran_hip_sch_create(srsran_hip_sch_t** hh)
{
  if (!hh) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  *hh = new srsran_hip_sch;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_sch_free(srsran_hip_sch_t* h)
{
  if (!h) {
    return;
  }
  for (auto& kv : h->dec) {
    srsran_hip_tdec_batch_free(kv.second.first);
  }
  (void)hipFree(h->d_scratch);
  (void)hipHostFree(h->h_scratch);
  (void)hipFree(h->d_crc_mult);
  (void)hipHostFree(h->h_crc_mult);
  delete h;
}

namespace {


srsran_hip_tdec_batch_t* decoder_for(srsran_hip_sch_t* h, uint32_t K, uint32_t n, bool llr8)
{
  const uint32_t key = K | (llr8 ? 0x80000000u : 0u);
  auto           it  = h->dec.find(key);
  if (it != h->dec.end() && it->second.second >= n) {
    return it->second.first;
  }
  if (it != h->dec.end()) {
    srsran_hip_tdec_batch_free(it->second.first);
    h->dec.erase(it);
  }
  srsran_hip_tdec_batch_t* b = nullptr;
  // (capacity: whatever the caller asks for, rounded up generously -- the object owns no memory, the workspace is the arena's)
  if (turbo::batch_create_shared(&b, K, n < 64 ? 64 : n, llr8, &h->arena)) {
    return nullptr;
  }
  n = n < 64 ? 64 : n;
  h->dec[key] = std::make_pair(b, n);
  return b;
}

} // namespace

// decode_tb (sch.c:507-572) for a batch; llr8 = q->llr_is_8bit: 8-bit rate de-matching and the 8-bit window decoders (:408-412,426-428)
// (tail: device-to-host copies a caller wants queued behind the last kernel and in front of the call's one host wait)
struct TailCopy {
  void*       dst;
  const void* src;
  size_t      bytes;
};
static int sch_decode(srsran_hip_sch_t* h, const void* d_e_bits, const srsran_hip_tb_t* tbs, uint32_t n_tb, uint32_t max_iterations, void* d_softbuf,
                      uint8_t* cb_crc, uint8_t* d_data, srsran_hip_tb_result_t* results, void* stream, bool llr8, const TailCopy* tail = nullptr,
                      int n_tail = 0, uint8_t* host_data = nullptr) // host_data: pinned mirror of d_data the transport-CRC kernel fills (tb_crc_kernel)
{
  TraceRange trace_("srsran_hip_sch_decode");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_sch_decode", SRSRAN_ERROR);
  }
  if (h && n_tb == 0) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!h || !d_e_bits || !tbs || !d_softbuf || !cb_crc || !d_data || !results || max_iterations == 0) {
    set_error("sch decode: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t st = (hipStream_t)stream;
  // The host's share of a call is two passes over the blocks with nothing allocated per code block: pass 1 segments and validates the
  // transport blocks and sizes the launch groups -- (K, rv, generator): one rate de-matching launch and one decoder launch each --, pass 2
  // writes every pending code block's job and descriptor straight into its place of the pinned image of the device arrays (32 k code
  // blocks per call in the benches: the work list, the per-group index vectors and their copies used to keep the GPU waiting 0.33 ms).
  struct TbPlan {
    srsran_cbsegm_t seg;
    int             grp[2] = {-1, -1}; // launch group of the blocks of size K1 / K2
    uint32_t        pend[2] = {0, 0};  // pending code blocks of each size (not decoded in an earlier HARQ round)
    uint32_t        start[2] = {0, 0}; // their first place in launch order
  };
  struct Group {
    uint64_t key;
    uint32_t K, rv, poly, count, base, cursor, max_in;
  };
  std::vector<TbPlan> plan(n_tb);
  std::vector<Group>  groups;
  size_t              n = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_hip_tb_t& tb = tbs[t];
    srsran_cbsegm_t&       cs = plan[t].seg;
    results[t] = {SRSRAN_ERROR, 0.f, 0};
    if (srsran_cbsegm(&cs, tb.tbs) || tb.Qm == 0 || (tb.rv & ~(uint32_t)SRSRAN_HIP_TB_NEW_DATA) > 3 || (tb.tbs & 7)) {
      set_error("sch decode: transport block %u: invalid tbs / Qm / rv", t);
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    results[t].nof_cb = cs.C;
    if (cs.tbs == 0 || cs.C == 0) {
      results[t].crc_ok = SRSRAN_SUCCESS; // sch.c:517-519
      continue;
    }
    if (cs.F) {
      fprintf(stderr, "Error filler bits are not supported. Use standard TBS\n"); // sch.c:521-524
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    // element / byte offsets travel as 32-bit values (rm::RxJob, turbo::CbDesc): refuse what would wrap instead of writing into other rows
    if ((uint64_t)(tb.first_cb + cs.C) * SRSRAN_HIP_SOFTBUFFER_CB_SIZE > 0xffffffffull || (uint64_t)tb.e_offset + tb.nof_e_bits > 0xffffffffull ||
        (uint64_t)tb.data_offset + tb.tbs / 8 + 3 > 0xffffffffull) {
      set_error("sch decode: transport block %u: input, soft-buffer or data offsets beyond 2^32", t);
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    for (uint32_t i = 0; i < cs.C; i++) {
      if (!cb_crc[tb.first_cb + i]) { // (else decoded in an earlier HARQ round: its bytes stay in d_data, sch.c:466-471)
        plan[t].pend[i < cs.C1 ? 0 : 1]++;
      }
    }
    for (int k = 0; k < 2; k++) {
      if (!plan[t].pend[k]) {
        continue;
      }
      const uint32_t K    = k == 0 ? cs.K1 : cs.K2;
      const uint32_t poly = cs.C > 1 ? CRC24B : CRC24A; // sch.c:432-438
      const uint64_t key  = ((uint64_t)K << 34) | ((uint64_t)(tb.rv & 3u) << 32) | poly;
      int            g    = -1;
      for (size_t j = 0; j < groups.size(); j++) {
        if (groups[j].key == key) {
          g = (int)j;
          break;
        }
      }
      if (g < 0) {
        g = (int)groups.size();
        groups.push_back({key, K, tb.rv & 3u, poly, 0, 0, 0, 0});
      }
      plan[t].grp[k] = g;
      groups[g].count += plan[t].pend[k];
      n += plan[t].pend[k];
    }
  }
  std::sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.key < b.key; });
  {
    // (the sort moved the groups: redo the blocks' group indices by key, then lay the groups out one behind the other)
    uint32_t base = 0;
    for (Group& g : groups) {
      g.base = g.cursor = base;
      base += g.count;
    }
    for (uint32_t t = 0; t < n_tb; t++) {
      const srsran_cbsegm_t& cs = plan[t].seg;
      for (int k = 0; k < 2; k++) {
        if (plan[t].grp[k] < 0) {
          continue;
        }
        const uint64_t key = ((uint64_t)(k == 0 ? cs.K1 : cs.K2) << 34) | ((uint64_t)(tbs[t].rv & 3u) << 32) | (cs.C > 1 ? CRC24B : CRC24A);
        for (size_t j = 0; j < groups.size(); j++) {
          if (groups[j].key == key) {
            plan[t].grp[k] = (int)j;
            break;
          }
        }
      }
    }
  }

  // Scratch, on the device and as a pinned host image of the same layout: per code block a job, a descriptor, an iteration count and a
  // verdict; per transport block a CRC job and its result.  [jobs | descriptors | CRC jobs] go up in one copy, [iteration counts | verdicts |
  // CRC results] come down in one, and the host waits ONCE: the transport-block CRC kernel decides by itself from the verdicts which blocks
  // are complete (sch.c:473-477), so nothing has to come back between decoding and it.
  auto         al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t o_jobs = 0, o_desc = al(o_jobs + n * sizeof(rm::RxJob)), o_tbj = al(o_desc + n * sizeof(turbo::CbDesc));
  const size_t o_noi = al(o_tbj + n_tb * sizeof(rm::TbCrcJob)), o_ok = al(o_noi + n * sizeof(int)), o_tbr = al(o_ok + n);
  const size_t bytes = al(o_tbr + n_tb * sizeof(rm::TbCrcResult));
  if (bytes > h->scratch_cap) {
    (void)hipFree(h->d_scratch);
    (void)hipHostFree(h->h_scratch);
    h->d_scratch = h->h_scratch = nullptr;
    h->scratch_cap               = 0;
    PHY_HIP_CHECK(hipMalloc(&h->d_scratch, bytes), SRSRAN_ERROR);
    PHY_HIP_CHECK(host_image_alloc(&h->h_scratch, bytes), SRSRAN_ERROR);
    h->scratch_cap = bytes;
  }
  uint8_t* hb   = static_cast<uint8_t*>(h->h_scratch);
  // A subframe's worth of code blocks or less: the kernels read the jobs / descriptors from, and write the verdicts into, the PINNED HOST image
  // themselves (it is mapped into the device's address space) -- four stream operations fewer per call (upload, clearing, two downloads;
  // 6-9 us each whatever the size, tools/probe/roundtrip_probe.hip), and what crosses the bus is a few hundred bytes.  Large batches keep the copies.
  const bool direct = n + n_tb <= 1024;
  uint8_t*   base   = direct ? hb : static_cast<uint8_t*>(h->d_scratch);
  auto*    d_jobs = reinterpret_cast<rm::RxJob*>(base + o_jobs);
  auto*    d_desc = reinterpret_cast<turbo::CbDesc*>(base + o_desc);
  auto*    d_tbj  = reinterpret_cast<rm::TbCrcJob*>(base + o_tbj);
  auto*    d_noi  = reinterpret_cast<int*>(base + o_noi);
  auto*    d_ok   = base + o_ok;
  auto*    d_tbr  = reinterpret_cast<rm::TbCrcResult*>(base + o_tbr);
  auto*    jobs   = reinterpret_cast<rm::RxJob*>(hb + o_jobs);
  auto*    descs  = reinterpret_cast<turbo::CbDesc*>(hb + o_desc);
  auto*    tbj    = reinterpret_cast<rm::TbCrcJob*>(hb + o_tbj);
  const int*             noi = reinterpret_cast<const int*>(hb + o_noi);
  const uint8_t*         ok  = hb + o_ok;
  const rm::TbCrcResult* tbr = reinterpret_cast<const rm::TbCrcResult*>(hb + o_tbr);

  // row of the CRC multiplier table for a block size: computed and uploaded the first time the size is seen by this object
  auto crc_row_of = [&](uint32_t tbs_bits) -> uint32_t {
    auto it = h->crc_row.find(tbs_bits);
    if (it != h->crc_row.end()) {
      return it->second;
    }
    const uint32_t row = (uint32_t)h->crc_row.size();
    if (row >= h->crc_rows_cap) {
      const uint32_t cap = h->crc_rows_cap ? 2 * h->crc_rows_cap : 16;
      uint32_t *     nd = nullptr, *nh = nullptr;
      if (hipMalloc(&nd, (size_t)cap * 256 * sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&nh, (size_t)cap * 256 * sizeof(uint32_t)) != hipSuccess ||
          hipDeviceSynchronize() != hipSuccess || // nothing may still read the old table / copy from the old mirror when they are freed below
          (h->d_crc_mult && (hipMemcpy(nd, h->d_crc_mult, (size_t)row * 256 * sizeof(uint32_t), hipMemcpyDeviceToDevice) != hipSuccess ||
                             hipDeviceSynchronize() != hipSuccess))) {
        (void)hipFree(nd);
        (void)hipHostFree(nh);
        set_error("sch decode: device allocation of the CRC multiplier table failed");
        return 0xffffffffu;
      }
      if (h->h_crc_mult) {
        memcpy(nh, h->h_crc_mult, (size_t)row * 256 * sizeof(uint32_t));
      }
      (void)hipFree(h->d_crc_mult);
      (void)hipHostFree(h->h_crc_mult);
      h->d_crc_mult   = nd;
      h->h_crc_mult   = nh;
      h->crc_rows_cap = cap;
    }
    uint32_t* m = h->h_crc_mult + (size_t)row * 256;
    rm::tb_crc_multipliers(tbs_bits, CRC24A, m);
    // (stream order puts the row in front of the CRC kernel of this call; the pinned source stays where it is)
    if (hipMemcpyAsync(h->d_crc_mult + (size_t)row * 256, m, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess) {
      set_error("sch decode: upload of the CRC multiplier table failed");
      return 0xffffffffu;
    }
    h->crc_row[tbs_bits] = row;
    return row;
  };
  // pass 2: jobs and descriptors in launch order; per transport block the CRC job with the two runs of verdicts that are its code blocks
  size_t n_crc = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_hip_tb_t& tb = tbs[t];
    const srsran_cbsegm_t& cs = plan[t].seg;
    if (cs.C == 0) {
      continue;
    }
    for (int k = 0; k < 2; k++) {
      if (plan[t].grp[k] >= 0) {
        plan[t].start[k] = groups[plan[t].grp[k]].cursor;
      }
    }
    // sch.c:389-405
    const uint32_t Gp    = tb.nof_e_bits / tb.Qm;
    const uint32_t gamma = Gp % cs.C;
    const uint32_t n_e   = tb.Qm * (Gp / cs.C);
    const uint32_t fresh = (tb.rv & SRSRAN_HIP_TB_NEW_DATA) ? 1u : 0u;
    for (uint32_t i = 0; i < cs.C; i++) {
      if (cb_crc[tb.first_cb + i]) {
        continue;
      }
      const int      k    = i < cs.C1 ? 0 : 1;
      const uint32_t K    = k == 0 ? cs.K1 : cs.K2;
      const uint32_t rlen = cs.C == 1 ? K : K - 24;
      uint32_t       rp = i * n_e, n_e2 = n_e;
      if (i > cs.C - gamma) {
        n_e2 = n_e + tb.Qm;
        rp   = (cs.C - gamma) * n_e + (i - (cs.C - gamma)) * n_e2;
      }
      Group&         g    = groups[plan[t].grp[k]];
      const uint32_t j    = g.cursor++;
      const uint32_t slot = tb.first_cb + i;
      jobs[j]  = {tb.e_offset + rp, n_e2, slot * (uint32_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE, 3 * K + 12, 0, fresh};
      descs[j] = {slot * (uint32_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE, tb.data_offset + i * rlen / 8, (i + 1 == cs.C) ? K / 8 : rlen / 8, 0};
      g.max_in = std::max(g.max_in, n_e2);
    }
    tbj[n_crc++] = {tb.data_offset, cs.tbs, {plan[t].start[0], plan[t].start[1]}, {plan[t].pend[0], plan[t].pend[1]}, plan[t].pend[0] + plan[t].pend[1],
                    crc_row_of(cs.tbs)};
    if (tbj[n_crc - 1].mult == 0xffffffffu) {
      return SRSRAN_ERROR;
    }
  }
  if (direct) {
    memset(hb + o_noi, 0, o_tbr - o_noi);
  } else if (n || n_crc) {
    PHY_HIP_CHECK(hipMemcpyAsync(base, hb, o_tbj + n_crc * sizeof(rm::TbCrcJob), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMemsetAsync(d_noi, 0, o_tbr - o_noi, st), SRSRAN_ERROR);
  }
  for (const Group& g : groups) {
    const uint32_t K = g.K, m = g.count, at = g.base;
    const uint32_t nsb = llr8 ? srsran_tdec_autoimp_get_subblocks_8bit(K) : srsran_tdec_autoimp_get_subblocks(K);
    const uint16_t* tab = rm::device_table(K, g.rv, nsb);
    srsran_hip_tdec_batch_t* dec = decoder_for(h, K, m, llr8);
    if (!tab || !dec) {
      return SRSRAN_ERROR;
    }
    // softbuffer += rate-matched soft bits (srsran_rm_turbo_rx_lut, sch.c:414), in the decoder's sub-block layout
    if ((size_t)g.max_in * (llr8 ? 1 : 2) > 40 * 1024 - 16) {
      // (the kernel for inputs beyond its LDS staging area only accumulates: clear the new blocks' rows here)
      for (uint32_t i = 0; i < m; i++) {
        const rm::RxJob& jb = jobs[at + i];
        if (jb.fresh) {
          PHY_HIP_CHECK(hipMemsetAsync(static_cast<uint8_t*>(d_softbuf) + (size_t)jb.out_offset * (llr8 ? 1 : 2), 0,
                                       (size_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE * (llr8 ? 1 : 2), st), SRSRAN_ERROR);
        }
      }
    }
    // (blocks of K <= 400 have no sub-block layout: nsb = 0, natural soft buffer, scalar decoder -- turbodecoder.c:381-408, rm_turbo.c:412-421)
    PHY_HIP_CHECK(rm::launch_rx_gather(d_e_bits, d_softbuf, tab, nsb ? 3 * (K + 32) + 12 : 3 * K + 12, d_jobs + at, rm::RxJob{}, 0, 0, (int)m, llr8, st, g.max_in), SRSRAN_ERROR);
    if (turbo::batch_run_early_stop(dec, d_softbuf, llr8, d_desc + at, d_data, m, max_iterations, nsb ? 1 : 0, g.poly, d_noi + at, d_ok + at, st)) {
      return SRSRAN_ERROR;
    }
  }
  // transport-block CRC (sch.c:473-477,540-560) of the blocks whose code blocks are all good, straight behind the decoders
  if (n_crc) {
    PHY_HIP_CHECK(rm::launch_tb_crc(d_data, d_tbj, (int)n_crc, CRC24A, d_ok, h->d_crc_mult, d_tbr, st, host_data), SRSRAN_ERROR);
  }
  if (!direct && (n || n_crc)) {
    PHY_HIP_CHECK(hipMemcpyAsync(hb + o_noi, base + o_noi, o_tbr + n_crc * sizeof(rm::TbCrcResult) - o_noi, hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
  }
  for (int i = 0; i < n_tail; i++) {
    if (tail[i].bytes) {
      PHY_HIP_CHECK(hipMemcpyAsync(tail[i].dst, tail[i].src, tail[i].bytes, hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    }
  }
  PHY_HIP_CHECK(hipStreamSynchronize(st), SRSRAN_ERROR);
  // results, walking the blocks in the order of pass 2
  size_t jc = 0;
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_hip_tb_t& tb = tbs[t];
    const srsran_cbsegm_t& cs = plan[t].seg;
    if (cs.C == 0) {
      continue;
    }
    float    it_sum = 0.f;
    uint32_t at[2]  = {plan[t].start[0], plan[t].start[1]};
    for (uint32_t i = 0; i < cs.C; i++) {
      if (cb_crc[tb.first_cb + i]) {
        continue;
      }
      const uint32_t j = at[i < cs.C1 ? 0 : 1]++;
      it_sum += (float)noi[j];
      if (ok[j]) {
        cb_crc[tb.first_cb + i] = 1;
      }
    }
    results[t].avg_iterations = it_sum / (float)cs.C; // sch.c:485
    if (tbr[jc].computed) {
      results[t].crc_ok = (tbr[jc].par_rx == tbr[jc].par_tx && tbr[jc].par_rx) ? SRSRAN_SUCCESS : SRSRAN_ERROR; // sch.c:551
    }
    jc++;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_sch_decode(srsran_hip_sch_t* h, const int16_t* d_e_bits, const srsran_hip_tb_t* tbs, uint32_t n_tb,
                                     uint32_t max_iterations, int16_t* d_softbuf, uint8_t* cb_crc, uint8_t* d_data,
                                     srsran_hip_tb_result_t* results, void* stream)
{
  return sch_decode(h, d_e_bits, tbs, n_tb, max_iterations, d_softbuf, cb_crc, d_data, results, stream, false);
}

extern "C" int srsran_hip_sch_decode_8bit(srsran_hip_sch_t* h, const int8_t* d_e_bits, const srsran_hip_tb_t* tbs, uint32_t n_tb,
                                          uint32_t max_iterations, int8_t* d_softbuf, uint8_t* cb_crc, uint8_t* d_data,
                                          srsran_hip_tb_result_t* results, void* stream)
{
  return sch_decode(h, d_e_bits, tbs, n_tb, max_iterations, d_softbuf, cb_crc, d_data, results, stream, true);
}

// ------------------------------------------------------------------------------------------------ the reference's transport-block seam
//
// decode_tb_cb (sch.c:370-492) on the caller's HOST buffers: what srsran_dlsch_decode2 / srsran_ulsch_decode reach through decode_tb
// (sch.c:541).  The staging context (stream, transport-block decoder, device and pinned buffers) is private to the calling thread, as
// the reference's worker threads each own their srsran_sch_t.

namespace {

struct TbStage {
  hipStream_t       st  = nullptr;
  srsran_hip_sch_t* sch = nullptr;
  uint8_t*          pin = nullptr; // pinned image: [soft rows | e bits | data]
  uint8_t*          dev = nullptr; // the same layout on the device
  size_t            cap = 0;
  bool              tried = false;
  ~TbStage()
  {
    srsran_hip_sch_free(sch);
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
        } else if (srsran_hip_sch_create(&sch) != SRSRAN_SUCCESS) {
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

// true when the n bytes at p (8-byte aligned, as the rows of a soft buffer are) are all zero: a row straight after srsran_softbuffer_rx_reset
// is not worth a copy, let alone a transfer.  Read-only, four independent accumulators (vectorises), early exit per 4 KB.
inline bool all_zero(const uint8_t* p, size_t n)
{
  size_t i = 0;
  if ((reinterpret_cast<uintptr_t>(p) & 7u) == 0) {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    const size_t    w = n / 8;
    for (size_t j = 0; j < w;) {
      const size_t e = j + 512 < w ? j + 512 : w;
      uint64_t     a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (; j + 4 <= e; j += 4) {
        a0 |= q[j], a1 |= q[j + 1], a2 |= q[j + 2], a3 |= q[j + 3];
      }
      for (; j < e; j++) {
        a0 |= q[j];
      }
      if (a0 | a1 | a2 | a3) {
        return false;
      }
    }
    i = w * 8;
  }
  for (; i < n; i++) {
    if (p[i]) {
      return false;
    }
  }
  return true;
}

} // namespace

namespace {
TbStage& tb_stage()
{
  static thread_local StageRef<TbStage> r;
  return r.get();
}
} // namespace

namespace phyhip {
namespace sch {
// the calling thread's transport-block stream (created on first use); nullptr without a device
hipStream_t stage_stream()
{
  TbStage& s = tb_stage();
  return s.ready() ? s.st : nullptr;
}
} // namespace sch
} // namespace phyhip

// decode_tb_cb for n transport blocks on the thread's stage: ONE upload, one de-matching and one early-stop decoder launch per block size over the code
// blocks of ALL of them, one transport-CRC launch, one download, one host wait (a second one only when a code block failed and its combined soft bits
// have to come back for the next transmission).  The e bits of a block are either the caller's host array (`e_bits`) or -- `front` given -- produced
// ON THE DEVICE by the kernels `front(stream, d_e_bits)` enqueues in front of the de-matcher (chan_host.cpp: equaliser, transform de-precoding,
// demodulator + descrambler never leave the device).  One call takes blocks of one soft-bit width, one iteration limit and one kind of e-bit source;
// a mixed list is decoded piece by piece.
static void tbs_staged_homogeneous(phyhip::sch::TbItem* it, uint32_t n, const phyhip::sch::GroupFrontEnd* group, uint32_t base)
{
  TraceRange trace_("decode_tb_cb (staged)");
  using phyhip::sch::TbItem;
  for (uint32_t t = 0; t < n; t++) {
    it[t].ok = false;
  }
  TbStage& s = tb_stage();
  if (!s.ready()) {
    fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: %s (there is no CPU fallback)\n", get_error());
    return;
  }
  auto*        q0   = static_cast<srsran_hip_sch_head_t*>(it[0].q);
  const bool   llr8 = q0->llr_is_8bit;
  const bool   dev_e = it[0].front != nullptr;
  const size_t es   = llr8 ? 1 : 2;
  const size_t row  = (size_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE * es;
  auto         al   = [](size_t v) { return (v + 255) & ~(size_t)255; };
  struct Plan {
    uint32_t C = 0, slot0 = 0;        // code blocks, first soft-buffer row of this block in the image
    size_t   o_e = 0, o_data = 0, n_data = 0;
    uint8_t  flags[32];
    uint32_t span[32], rbytes[32];
    bool     any_flag = false, any_soft = false, live = false; // live: takes part in the launch
    int      first = -1, last = -1;
  };
  std::vector<Plan> pl(n);
  uint32_t          slots = 0;
  for (uint32_t t = 0; t < n; t++) {
    TbItem& x = it[t];
    auto*   q = static_cast<srsran_hip_sch_head_t*>(x.q);
    Plan&   p = pl[t];
    const uint32_t C = x.seg->C;
    q->avg_iterations = 0; // sch.c:387
    if (C == 0) { // no code block: the loops of sch.c:389-486 do not run (0 / 0 is what the reference leaves in avg_iterations, too)
      x.sb->tb_crc      = true;
      q->avg_iterations = 0.f / 0.f;
      x.ok              = true;
      continue;
    }
    if (C > x.sb->max_cb) {
      continue;
    }
    {
      // the kernels' work lists are derived from the transport-block size: a segmentation that is not THE segmentation of cb_segm->tbs (a caller's
      // hand-made struct) would index the per-block arrays below with two different block counts
      srsran_cbsegm_t chk;
      if (srsran_cbsegm(&chk, x.seg->tbs) != SRSRAN_SUCCESS || chk.C != C || chk.C1 != x.seg->C1 || chk.K1 != x.seg->K1 ||
          (chk.C2 && chk.K2 != x.seg->K2) || chk.C2 != x.seg->C2) {
        fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: cb_segm is not srsran_cbsegm(tbs = %u)\n", x.seg->tbs);
        continue;
      }
    }
    p.C     = C;
    p.slot0 = slots;
    slots += C;
    p.live = true;
  }
  size_t off = al((size_t)slots * row);
  for (uint32_t t = 0; t < n; t++) {
    if (pl[t].live) {
      pl[t].o_e = off;
      off       = al(off + (size_t)it[t].nof_e_bits * es);
    }
  }
  const size_t o_data0 = off;
  for (uint32_t t = 0; t < n; t++) {
    if (pl[t].live) {
      pl[t].o_data = off;
      pl[t].n_data = it[t].seg->tbs / 8 + 6; // the last block's K/8 bytes end 3 bytes behind the transport CRC (:424 writes whole blocks)
      off          = al(off + pl[t].n_data);
    }
  }
  const size_t total = off;
  if (slots == 0) {
    return;
  }
  if (!s.grow(total)) {
    fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: staging allocation failed\n");
    return;
  }
  std::vector<srsran_hip_tb_t>        tbs;
  std::vector<srsran_hip_tb_result_t> res;
  std::vector<uint32_t>               who; // item of each launched block
  std::vector<uint8_t>                cbflags(slots, 1);
  bool                                any_data_up = false;
  for (uint32_t t = 0; t < n; t++) {
    Plan& p = pl[t];
    if (!p.live) {
      continue;
    }
    TbItem&                 x  = it[t];
    srsran_softbuffer_rx_t* sb = x.sb;
    const size_t            o_soft = (size_t)p.slot0 * row;
    // per code block: size, payload bytes, soft-buffer span (the layout srsran_rm_turbo_rx_lut{,_8bit} fills for the AUTO decoder of that size)
    for (uint32_t i = 0; i < p.C; i++) {
      const uint32_t K   = i < x.seg->C1 ? x.seg->K1 : x.seg->K2;
      const uint32_t nsb = llr8 ? srsran_tdec_autoimp_get_subblocks_8bit(K) : srsran_tdec_autoimp_get_subblocks(K);
      p.span[i]   = nsb ? 3 * (K + 32) + 12 : 3 * K + 12;
      p.rbytes[i] = (p.C == 1 ? K : K - 24) / 8;
      p.flags[i]  = sb->cb_crc[i] ? 1 : 0;
      if (p.flags[i]) {
        p.any_flag = true;
        // decoded in an earlier round: its stored bytes (sch.c:466-471), which the transport CRC on the device needs too
        memcpy(s.pin + p.o_data + (size_t)i * p.rbytes[i], sb->data[i], p.rbytes[i]);
      } else {
        p.any_soft = p.any_soft || !all_zero(reinterpret_cast<const uint8_t*>(sb->buffer_f[i]), p.span[i] * es);
        p.first    = p.first < 0 ? (int)i : p.first;
        p.last     = (int)i;
      }
    }
    if (p.first < 0) {
      // every code block was decoded in an earlier round (not a state the reference's callers produce: they reset a decoded block's soft buffer):
      // nothing to launch, the stored bytes are the block (sch.c:466-471), no iteration is counted
      for (uint32_t i = 0; i < p.C; i++) {
        memcpy(&x.data[(size_t)i * p.rbytes[i]], sb->data[i], p.rbytes[i]);
      }
      sb->tb_crc = true;
      x.ok       = true;
      p.live     = false;
      continue;
    }
    uint32_t rv = x.rv;
    if (p.any_soft || p.any_flag) { // a retransmission: the rows hold the earlier transmissions' soft bits
      for (uint32_t i = 0; i < p.C; i++) {
        if (!p.flags[i]) {
          memcpy(s.pin + o_soft + i * row, sb->buffer_f[i], p.span[i] * es);
        }
      }
      if (hipMemcpyAsync(s.dev + o_soft + p.first * row, s.pin + o_soft + p.first * row, (p.last - p.first) * row + p.span[p.last] * es,
                         hipMemcpyHostToDevice, s.st) != hipSuccess) {
        (void)hipStreamSynchronize(s.st);
        return;
      }
    } else {
      rv |= SRSRAN_HIP_TB_NEW_DATA; // every row is still zero: the de-matcher writes the rows instead of accumulating into them
    }
    any_data_up = any_data_up || p.any_flag;
    if (!dev_e) {
      memcpy(s.pin + p.o_e, x.e_bits, (size_t)x.nof_e_bits * es); // (the de-matcher reads them ONCE, coalesced: straight from the pinned image, no copy operation)
    }
    for (uint32_t i = 0; i < p.C; i++) {
      cbflags[p.slot0 + i] = p.flags[i];
    }
    // offsets in elements of the e-bit type from the image's start, data offsets in bytes from the data region's start
    tbs.push_back({x.seg->tbs, x.Qm, rv, x.nof_e_bits, (uint32_t)(p.o_e / es), (uint32_t)(p.o_data - o_data0), p.slot0});
    res.push_back({SRSRAN_ERROR, 0.f, 0});
    who.push_back(t);
  }
  if (tbs.empty()) {
    return;
  }
  if (any_data_up && hipMemcpyAsync(s.dev + o_data0, s.pin + o_data0, total - o_data0, hipMemcpyHostToDevice, s.st) != hipSuccess) {
    (void)hipStreamSynchronize(s.st);
    return;
  }
  if (dev_e && group) {
    std::vector<uint32_t> idx;
    std::vector<void*>    de;
    for (uint32_t t : who) {
      idx.push_back(base + t);
      de.push_back(s.dev + pl[t].o_e);
    }
    if (!(*group)(s.st, idx.data(), de.data(), (uint32_t)idx.size())) {
      (void)hipStreamSynchronize(s.st);
      fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: %s\n", get_error());
      return;
    }
  } else if (dev_e) {
    for (uint32_t t : who) {
      if (!(*it[t].front)(s.st, s.dev + pl[t].o_e)) {
        (void)hipStreamSynchronize(s.st); // nothing of a failed call may still be in flight when the next one re-uses the images
        fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: %s\n", get_error());
        return;
      }
    }
  }
  // the decoded bytes come back in front of the call's one host wait (the transport-CRC kernel writes them into the pinned image); the combined soft bits
  // only when a block failed (second wait, below)
  const int rc = sch_decode(s.sch, dev_e ? s.dev : s.pin, tbs.data(), (uint32_t)tbs.size(), q0->max_iterations ? q0->max_iterations : 1, s.dev, cbflags.data(),
                            s.dev + o_data0, res.data(), s.st, llr8, nullptr, 0, s.pin + o_data0);
  if (rc != SRSRAN_SUCCESS) {
    (void)hipStreamSynchronize(s.st);
    fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: %s\n", get_error());
    return;
  }
  // still undecoded: their rows are the HARQ state the next transmission combines into
  bool second = false;
  for (uint32_t t : who) {
    Plan& p = pl[t];
    int   f_first = -1, f_last = -1;
    for (uint32_t i = 0; i < p.C; i++) {
      if (!cbflags[p.slot0 + i]) {
        f_first = f_first < 0 ? (int)i : f_first;
        f_last  = (int)i;
      }
    }
    if (f_first >= 0) {
      const size_t o_soft = (size_t)p.slot0 * row;
      if (hipMemcpyAsync(s.pin + o_soft + f_first * row, s.dev + o_soft + f_first * row, (f_last - f_first) * row + p.span[f_last] * es,
                         hipMemcpyDeviceToHost, s.st) != hipSuccess) {
        (void)hipStreamSynchronize(s.st);
        fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: download of the soft buffer rows failed\n");
        return;
      }
      second = true;
    }
  }
  if (second && hipStreamSynchronize(s.st) != hipSuccess) {
    fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: download of the soft buffer rows failed\n");
    return;
  }
  // host side effects of sch.c:424-486
  for (size_t k = 0; k < who.size(); k++) {
    const uint32_t          t  = who[k];
    Plan&                   p  = pl[t];
    TbItem&                 x  = it[t];
    srsran_softbuffer_rx_t* sb = x.sb;
    const size_t            o_soft = (size_t)p.slot0 * row;
    bool                    all_ok = true;
    for (uint32_t i = 0; i < p.C; i++) {
      if (sb->cb_crc[i]) {
        memcpy(&x.data[(size_t)i * p.rbytes[i]], sb->data[i], p.rbytes[i]);
        continue;
      }
      const uint32_t K = i < x.seg->C1 ? x.seg->K1 : x.seg->K2;
      memcpy(&x.data[(size_t)i * p.rbytes[i]], s.pin + p.o_data + (size_t)i * p.rbytes[i], i + 1 == p.C ? K / 8 : p.rbytes[i]);
      if (cbflags[p.slot0 + i]) {
        sb->cb_crc[i] = true;
      } else {
        all_ok = false;
        // still undecoded: the next transmission combines into this row
        memcpy(sb->buffer_f[i], s.pin + o_soft + i * row, p.span[i] * es);
      }
    }
    sb->tb_crc = all_ok;
    if (!all_ok) {
      for (uint32_t i = 0; i < p.C; i++) {
        if (sb->cb_crc[i]) {
          memcpy(sb->data[i], &x.data[(size_t)i * p.rbytes[i]], p.rbytes[i]); // sch.c:476-484
        }
      }
    }
    static_cast<srsran_hip_sch_head_t*>(x.q)->avg_iterations = res[k].avg_iterations;
    x.ok = all_ok;
  }
}

void phyhip::sch::decode_tbs_staged(TbItem* it, uint32_t n, const GroupFrontEnd* group)
{
  // validate, then cut the list into runs of blocks that can share a launch
  for (uint32_t t = 0; t < n; t++) {
    TbItem& x = it[t];
    x.ok      = false;
    if (!x.q || !x.sb || !x.seg || (!x.e_bits && !x.front) || !x.data || x.Qm == 0 || x.rv > 3) {
      fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: invalid arguments\n");
      x.seg = nullptr; // skipped below
    } else if (x.seg->C > 32) { // SRSRAN_MAX_CODEBLOCKS, sch.c:382-385
      fprintf(stderr, "Error SRSRAN_MAX_CODEBLOCKS=%d\n", 32);
      x.seg = nullptr;
    }
  }
  uint32_t a = 0;
  while (a < n) {
    if (!it[a].seg) {
      a++;
      continue;
    }
    auto*    qa = static_cast<srsran_hip_sch_head_t*>(it[a].q);
    uint32_t b  = a + 1;
    while (b < n && it[b].seg) {
      auto* qb = static_cast<srsran_hip_sch_head_t*>(it[b].q);
      if (qb->llr_is_8bit != qa->llr_is_8bit || qb->max_iterations != qa->max_iterations || (it[b].front != nullptr) != (it[a].front != nullptr)) {
        break;
      }
      b++;
    }
    tbs_staged_homogeneous(it + a, b - a, group, a);
    a = b;
  }
}

bool phyhip::sch::decode_tb_staged(void* qv, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits,
                                   const void* e_bits, const FrontEnd* front, uint8_t* data)
{
  TbItem x = {qv, softbuffer, cb_segm, Qm, rv, nof_e_bits, e_bits, front, data, false};
  decode_tbs_staged(&x, 1);
  return x.ok;
}

extern "C" bool srsran_hip_decode_tb_cb(void* qv, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv,
                                        uint32_t nof_e_bits, void* e_bits, uint8_t* data)
{
  if (!e_bits) {
    fprintf(stderr, "[srsran_phy_hip] decode_tb_cb: invalid arguments\n");
    return false;
  }
  return phyhip::sch::decode_tb_staged(qv, softbuffer, cb_segm, Qm, rv, nof_e_bits, e_bits, nullptr, data);
}

extern "C" bool decode_tb_cb(void* q, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits,
                             void* e_bits, uint8_t* data)
{
  return srsran_hip_decode_tb_cb(q, softbuffer, cb_segm, Qm, rv, nof_e_bits, e_bits, data);
}
