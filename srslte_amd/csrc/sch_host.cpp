// sch_host.cpp -- transport-block decoding on the device: srsran_cbsegm, and decode_tb / decode_tb_cb of
// lib/src/phy/phch/sch.c:370-560 for a batch of transport blocks (rate de-matching -> turbo half iterations with
// CRC early stop per code block -> transport-block CRC).
#include "hip_common.h"
#include "rm_device.h"
#include "srsran_amd/phy_sch_abi.h"
#include "turbo_device.h"

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
  // one turbo batch object per (K, arithmetic is 16 bit), grown on demand
  std::map<uint32_t, std::pair<srsran_hip_tdec_batch_t*, uint32_t>> dec; // K | 8-bit flag << 31 -> (object, capacity)
  void*  d_scratch = nullptr; // job / descriptor / result arrays
  size_t scratch_cap = 0;
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
  delete h;
}

namespace {

struct CbWork {
  uint32_t tb, cb_idx, K, slot, poly;
  rm::RxJob       job;
  turbo::CbDesc   desc;
};

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
  if (llr8 ? srsran_hip_tdec_batch_create_8bit(&b, K, n, SRSRAN_TDEC_AUTO) : srsran_hip_tdec_batch_create(&b, K, n, SRSRAN_TDEC_AUTO)) {
    return nullptr;
  }
  h->dec[key] = std::make_pair(b, n);
  return b;
}

} // namespace

// decode_tb (sch.c:507-572) for a batch; llr8 = q->llr_is_8bit: 8-bit rate de-matching and the 8-bit window decoders (:408-412,426-428)
static int sch_decode(srsran_hip_sch_t* h, const void* d_e_bits, const srsran_hip_tb_t* tbs, uint32_t n_tb, uint32_t max_iterations, void* d_softbuf,
                      uint8_t* cb_crc, uint8_t* d_data, srsran_hip_tb_result_t* results, void* stream, bool llr8)
{
  if (h && n_tb == 0) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!h || !d_e_bits || !tbs || !d_softbuf || !cb_crc || !d_data || !results || max_iterations == 0) {
    set_error("sch decode: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  hipStream_t          st = (hipStream_t)stream;
  std::vector<CbWork>  work;
  std::vector<srsran_cbsegm_t> seg(n_tb);
  std::vector<float>   iters(n_tb, 0.f);
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_hip_tb_t& tb = tbs[t];
    results[t] = {SRSRAN_ERROR, 0.f, 0};
    if (srsran_cbsegm(&seg[t], tb.tbs) || tb.Qm == 0 || (tb.rv & ~(uint32_t)SRSRAN_HIP_TB_NEW_DATA) > 3 || (tb.tbs & 7)) {
      set_error("sch decode: transport block %u: invalid tbs / Qm / rv", t);
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    const srsran_cbsegm_t& cs = seg[t];
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
      if (cb_crc[tb.first_cb + i]) {
        continue; // decoded in an earlier HARQ round: its bytes stay in d_data (sch.c:466-471)
      }
      // sch.c:389-405
      const uint32_t K     = i < cs.C1 ? cs.K1 : cs.K2;
      const uint32_t rlen  = cs.C == 1 ? K : K - 24;
      const uint32_t Gp    = tb.nof_e_bits / tb.Qm;
      const uint32_t gamma = Gp % cs.C;
      const uint32_t n_e   = tb.Qm * (Gp / cs.C);
      uint32_t       rp = i * n_e, n_e2 = n_e;
      if (i > cs.C - gamma) {
        n_e2 = n_e + tb.Qm;
        rp   = (cs.C - gamma) * n_e + (i - (cs.C - gamma)) * n_e2;
      }
      CbWork w;
      w.tb     = t;
      w.cb_idx = i;
      w.K      = K;
      w.slot   = tb.first_cb + i;
      w.poly   = cs.C > 1 ? CRC24B : CRC24A; // sch.c:432-438
      w.job    = {tb.e_offset + rp, n_e2, w.slot * (uint32_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE, 3 * K + 12, 0, (tb.rv & SRSRAN_HIP_TB_NEW_DATA) ? 1u : 0u};
      w.desc   = {w.slot * (uint32_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE, tb.data_offset + i * rlen / 8,
                  (i + 1 == cs.C) ? K / 8 : rlen / 8, 0};
      work.push_back(w);
    }
  }
  // device scratch: per code block a job, a descriptor, an iteration count and a flag; per block a CRC job + result
  const size_t n = work.size();
  const size_t bytes = n * (sizeof(rm::RxJob) + sizeof(turbo::CbDesc) + sizeof(int) + 4) + n_tb * (sizeof(rm::TbCrcJob) + sizeof(rm::TbCrcResult)) + 256;
  if (bytes > h->scratch_cap) {
    (void)hipFree(h->d_scratch);
    h->d_scratch = nullptr;
    PHY_HIP_CHECK(hipMalloc(&h->d_scratch, bytes), SRSRAN_ERROR);
    h->scratch_cap = bytes;
  }
  uint8_t* base = static_cast<uint8_t*>(h->d_scratch);
  auto*    d_jobs = reinterpret_cast<rm::RxJob*>(base);
  auto*    d_desc = reinterpret_cast<turbo::CbDesc*>(d_jobs + n);
  auto*    d_noi  = reinterpret_cast<int*>(d_desc + n);
  auto*    d_ok   = reinterpret_cast<uint8_t*>(d_noi + n);
  auto*    d_tbj  = reinterpret_cast<rm::TbCrcJob*>(base + ((reinterpret_cast<uintptr_t>(d_ok + n) - reinterpret_cast<uintptr_t>(base) + 15) & ~(uintptr_t)15));
  auto*    d_tbr  = reinterpret_cast<rm::TbCrcResult*>(d_tbj + n_tb);

  // group by (K, rv, generator): one rate de-matching launch and one decoder launch per group
  std::map<uint64_t, std::vector<size_t>> groups;
  for (size_t i = 0; i < n; i++) {
    groups[((uint64_t)work[i].K << 34) | ((uint64_t)(tbs[work[i].tb].rv & 3u) << 32) | work[i].poly].push_back(i);
  }
  std::vector<rm::RxJob>     jobs;
  std::vector<turbo::CbDesc> descs;
  std::vector<size_t>        order;
  for (auto& g : groups) {
    for (size_t i : g.second) {
      jobs.push_back(work[i].job);
      descs.push_back(work[i].desc);
      order.push_back(i);
    }
  }
  if (n) {
    PHY_HIP_CHECK(hipMemcpyAsync(d_jobs, jobs.data(), n * sizeof(rm::RxJob), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMemcpyAsync(d_desc, descs.data(), n * sizeof(turbo::CbDesc), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMemsetAsync(d_noi, 0, n * sizeof(int) + n, st), SRSRAN_ERROR);
  }
  size_t at = 0;
  for (auto& g : groups) {
    const uint32_t K = (uint32_t)(g.first >> 34), rv = (uint32_t)((g.first >> 32) & 3), poly = (uint32_t)g.first;
    const uint32_t m = (uint32_t)g.second.size();
    const uint32_t nsb = llr8 ? srsran_tdec_autoimp_get_subblocks_8bit(K) : srsran_tdec_autoimp_get_subblocks(K);
    const uint16_t* tab = rm::device_table(K, rv, nsb);
    srsran_hip_tdec_batch_t* dec = decoder_for(h, K, m, llr8);
    if (!tab || !dec) {
      return SRSRAN_ERROR;
    }
    // softbuffer += rate-matched soft bits (srsran_rm_turbo_rx_lut, sch.c:414), in the decoder's sub-block layout
    uint32_t max_in = 0;
    for (uint32_t i = 0; i < m; i++) {
      max_in = std::max(max_in, work[g.second[i]].job.in_len);
    }
    if ((size_t)max_in * (llr8 ? 1 : 2) > 40 * 1024 - 16) {
      // (the kernel for inputs beyond its LDS staging area only accumulates: clear the new blocks' rows here)
      for (uint32_t i = 0; i < m; i++) {
        const CbWork& w = work[g.second[i]];
        if (w.job.fresh) {
          PHY_HIP_CHECK(hipMemsetAsync(static_cast<uint8_t*>(d_softbuf) + (size_t)w.job.out_offset * (llr8 ? 1 : 2), 0,
                                       (size_t)SRSRAN_HIP_SOFTBUFFER_CB_SIZE * (llr8 ? 1 : 2), st), SRSRAN_ERROR);
        }
      }
    }
    // (blocks of K <= 400 have no sub-block layout: nsb = 0, natural soft buffer, scalar decoder -- turbodecoder.c:381-408, rm_turbo.c:412-421)
    PHY_HIP_CHECK(rm::launch_rx_gather(d_e_bits, d_softbuf, tab, nsb ? 3 * (K + 32) + 12 : 3 * K + 12, d_jobs + at, rm::RxJob{}, 0, 0, (int)m, llr8, st, max_in), SRSRAN_ERROR);
    if (turbo::batch_run_early_stop(dec, d_softbuf, llr8, d_desc + at, d_data, m, max_iterations, nsb ? 1 : 0, poly, d_noi + at, d_ok + at, st)) {
      return SRSRAN_ERROR;
    }
    at += m;
  }
  std::vector<int>     noi(n);
  std::vector<uint8_t> ok(n);
  if (n) {
    PHY_HIP_CHECK(hipMemcpyAsync(noi.data(), d_noi, n * sizeof(int), hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMemcpyAsync(ok.data(), d_ok, n, hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
  }
  PHY_HIP_CHECK(hipStreamSynchronize(st), SRSRAN_ERROR);
  for (size_t j = 0; j < n; j++) {
    const CbWork& w = work[order[j]];
    if (ok[j]) {
      cb_crc[w.slot] = 1;
    }
    iters[w.tb] += (float)noi[j];
  }
  // transport-block CRC of the blocks whose code blocks are all good (sch.c:473-477,540-560)
  std::vector<rm::TbCrcJob> tbj;
  std::vector<uint32_t>     tbi;
  for (uint32_t t = 0; t < n_tb; t++) {
    const srsran_cbsegm_t& cs = seg[t];
    if (cs.C == 0) {
      continue;
    }
    results[t].avg_iterations = iters[t] / (float)cs.C; // sch.c:485
    bool all = true;
    for (uint32_t i = 0; i < cs.C; i++) {
      all = all && cb_crc[tbs[t].first_cb + i];
    }
    if (all) {
      tbj.push_back({tbs[t].data_offset, cs.tbs});
      tbi.push_back(t);
    }
  }
  if (!tbj.empty()) {
    std::vector<rm::TbCrcResult> r(tbj.size());
    PHY_HIP_CHECK(hipMemcpyAsync(d_tbj, tbj.data(), tbj.size() * sizeof(rm::TbCrcJob), hipMemcpyHostToDevice, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(rm::launch_tb_crc(d_data, d_tbj, (int)tbj.size(), CRC24A, d_tbr, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMemcpyAsync(r.data(), d_tbr, r.size() * sizeof(rm::TbCrcResult), hipMemcpyDeviceToHost, st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipStreamSynchronize(st), SRSRAN_ERROR);
    for (size_t j = 0; j < r.size(); j++) {
      results[tbi[j]].crc_ok = (r[j].par_rx == r[j].par_tx && r[j].par_rx) ? SRSRAN_SUCCESS : SRSRAN_ERROR; // sch.c:551
    }
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
