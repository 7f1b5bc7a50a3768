// ldpc_host.cpp -- host side of the NR LDPC decoder: base graphs, batch object, srsran_ldpc_decoder_* ABI.
//
// Mirrors (interface + behaviour) lib/src/phy/fec/ldpc/ldpc_decoder.c:540-685 and
// lib/src/phy/fec/ldpc/base_graph.c:50,4467-4503 of the reference.
#include "coalesce.h"
#include "hip_common.h"
#include "ldpc_device.h"
#include "tables/nr_ldpc_bg_table.h"
#include "tables/nr_ldpc_lsindex.h"

#include <atomic>
#include <algorithm>
#include <map>
#include <vector>

using namespace phyhip;

// ------------------------------------------------------------------------------------------------ base graph

static int ls_index_of(unsigned ls)
{
  // TS 38.212 Table 5.3.2-1: Z = a * 2^j, a in {2,3,5,7,9,11,13,15} (set index 0..7), 2 <= Z <= 384
  if (ls < 2 || ls > MAX_LIFTSIZE) {
    return VOID_LIFTSIZE;
  }
  unsigned odd = ls;
  while (!(odd & 1)) {
    odd >>= 1;
  }
  static const int set_of_odd[16] = {-1, 0, -1, 1, -1, 2, -1, 3, -1, 4, -1, 5, -1, 6, -1, 7};
  if (odd > 15 || set_of_odd[odd] < 0) {
    return VOID_LIFTSIZE;
  }
  return set_of_odd[odd];
}

// base_graph.c:50 -- exported under the reference's name
extern "C" const uint8_t LSindex[385] = NR_LDPC_LSINDEX_INIT;

struct BgDims {
  int N, M, K, E;
  const nr_ldpc_edge_t* edges;
};

static bool bg_dims(int bg, BgDims* d)
{
  if (bg == BG1) {
    *d = {68, 46, 22, NR_LDPC_BG1_NOF_EDGES, nr_ldpc_bg1_edges};
    return true;
  }
  if (bg == BG2) {
    *d = {52, 42, 10, NR_LDPC_BG2_NOF_EDGES, nr_ldpc_bg2_edges};
    return true;
  }
  return false;
}

extern "C" int create_compact_pcm(uint16_t* pcm, int8_t (*positions)[MAX_CNCT], srsran_basegraph_t bg, uint16_t ls)
{
  BgDims d;
  int    ils = ls_index_of(ls);
  if (!bg_dims(bg, &d) || ils == VOID_LIFTSIZE) {
    fprintf(stderr, "Invalid lifting size %d\n", ls);
    return -1;
  }
  for (int i = 0; i < d.M * d.N; i++) {
    pcm[i] = NO_CNCT;
  }
  if (positions) {
    for (int m = 0; m < d.M; m++) {
      for (int j = 0; j < MAX_CNCT; j++) {
        positions[m][j] = -1;
      }
    }
  }
  std::vector<int> fill(d.M, 0);
  for (int e = 0; e < d.E; e++) {
    const nr_ldpc_edge_t& ed = d.edges[e];
    pcm[ed.row * d.N + ed.col] = (uint16_t)(ed.v[ils] % ls);
    if (positions) {
      positions[ed.row][fill[ed.row]++] = (int8_t)ed.col;
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ batch object

// Code words per workgroup.  Small lifting sizes share a 256-lane workgroup.  Large ones are balanced over the four SIMDs of
// a CU: a 6-wave workgroup (Z = 384) loads them 2 + 2 + 1 + 1, and whether a second workgroup complements that is up to the
// dispatcher; two code words in one 12-wave workgroup are 3 + 3 + 3 + 3 by construction (measured 50.4 -> 44.3 ms).  The
// score is code words per busiest-SIMD wave; ties go to the smaller workgroup.
static int choose_cpb(int Z, size_t soft_bytes_per_cw)
{
  if (Z <= 128) {
    int cpb = 256 / Z;
    while (cpb > 1 && (size_t)cpb * soft_bytes_per_cw > 62 * 1024) {
      cpb--;
    }
    return cpb;
  }
  int    best = 1;
  double best_score = 0;
  for (int cpb = 1; cpb * Z <= 768; cpb++) {
    if ((size_t)cpb * soft_bytes_per_cw > 150 * 1024) {
      break;
    }
    const int    waves = (cpb * Z + 63) / 64;
    const double score = (double)cpb / (double)((waves + 3) / 4);
    if (score > best_score + 1e-9) {
      best_score = score;
      best       = cpb;
    }
  }
  return best;
}


struct srsran_hip_ldpc_batch {
  DeviceTag tag;
  int      bg = 0, Z = 0, N = 0, M = 0, K = 0, E = 0;
  int      max_iter = 0;
  int      sf       = 0;
  float    sf_f     = 0.f;
  int      dtype    = ldpc::DT_I8; // message type: int8 (ldpc_dec_c.c), int16 (ldpc_dec_s.c) or float (ldpc_dec_f.c)
  bool     flood    = false;       // flooded schedule (ldpc_dec_c_flood.c), int8 only
  int      cpb      = 1;           // code words per workgroup (choose_cpb)
  int      slots    = 1;           // c2v slabs allocated = most workgroups a launch may use
  int*     d_col_start = nullptr;
  int*     d_col_edges = nullptr;
  // x^((Z-1-c) bgK) mod g per generator used with this object (CRC early stop): one device table each, filled on first use and kept --
  // a batch that alternates CRC24A / CRC24B / CRC16 code words (sch_nr.c:606-619) never re-uploads or synchronises
  std::map<uint64_t, uint32_t*> crc_mult;
  void*    d_c2v    = nullptr;     // check-to-variable messages: slots x cpb slabs of E x Z messages
  unsigned int* d_work = nullptr;  // packed kernel: the counter its workgroups take their code words from
  uint32_t max_cw   = 0;
  std::vector<uint16_t> row_start;
  int* d_row_start = nullptr;
  int* d_edges     = nullptr;
};

extern "C" int srsran_hip_ldpc_batch_create(srsran_hip_ldpc_batch_t** hh, srsran_basegraph_t bg, uint16_t ls,
                                            float scaling_fctr, uint32_t max_nof_iter, uint32_t max_nof_cw)
{
  return srsran_hip_ldpc_batch_create_typed(hh, bg, ls, scaling_fctr, max_nof_iter, max_nof_cw, SRSRAN_LDPC_DECODER_C);
}

extern "C" int srsran_hip_ldpc_batch_create_typed(srsran_hip_ldpc_batch_t** hh, srsran_basegraph_t bg, uint16_t ls,
                                                  float scaling_fctr, uint32_t max_nof_iter, uint32_t max_nof_cw,
                                                  srsran_ldpc_decoder_type_t type)
{
  if (!hh) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  int  dtype;
  bool flood = false;
  switch (type) {
    case SRSRAN_LDPC_DECODER_F:
      dtype = ldpc::DT_F32;
      break;
    case SRSRAN_LDPC_DECODER_S:
      dtype = ldpc::DT_I16;
      break;
    case SRSRAN_LDPC_DECODER_C:
    case SRSRAN_LDPC_DECODER_C_AVX2:
    case SRSRAN_LDPC_DECODER_C_AVX512:
      dtype = ldpc::DT_I8; // one family: identical results in the reference
      break;
    case SRSRAN_LDPC_DECODER_C_FLOOD:
    case SRSRAN_LDPC_DECODER_C_AVX2_FLOOD:
    case SRSRAN_LDPC_DECODER_C_AVX512_FLOOD:
      // the flooded schedule, as the scalar ldpc_dec_c_flood.c computes it (the reference's AVX2 "long" flooded decoder
      // differs from its own scalar one on shortened code words; the scalar one is followed)
      dtype = ldpc::DT_I8;
      flood = true;
      break;
    default:
      set_error("LDPC decoder type %d does not exist", (int)type);
      return SRSRAN_ERROR_INVALID_INPUTS;
  }
  *hh = nullptr;
  BgDims d;
  int    ils = ls_index_of(ls);
  if (!bg_dims(bg, &d) || ils == VOID_LIFTSIZE) {
    set_error("invalid base graph %d / lifting size %d", bg, ls);
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (scaling_fctr <= 0 || scaling_fctr > 1) { // ldpc_decoder.c:601
    set_error("The scaling factor of the min-sum algorithm should be larger than 0 and not larger than 1.");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  if (!device_available()) {
    return SRSRAN_ERROR;
  }
  auto* h     = new srsran_hip_ldpc_batch;
  h->bg       = bg;
  h->Z        = ls;
  h->N        = d.N;
  h->M        = d.M;
  h->K        = d.K;
  h->E        = d.E;
  h->max_iter = max_nof_iter ? (int)max_nof_iter : 10; // ldpc_decoder.c:42,579
  h->sf       = (int)(scaling_fctr * 100);             // ldpc_dec_c.c:150 (float * int, truncated)
  h->sf_f     = scaling_fctr;
  h->dtype    = dtype;
  h->flood    = flood;
  h->max_cw   = max_nof_cw;
  std::vector<uint8_t>  col(d.E);
  std::vector<uint16_t> shift(d.E);
  h->row_start.assign(d.M + 1, 0);
  int row = 0;
  for (int e = 0; e < d.E; e++) {
    while (row < d.edges[e].row) {
      h->row_start[++row] = (uint16_t)e;
    }
    col[e]   = d.edges[e].col;
    shift[e] = (uint16_t)(d.edges[e].v[ils] % ls);
  }
  while (row < d.M) {
    h->row_start[++row] = (uint16_t)d.E;
  }
  std::vector<int> rs(h->row_start.begin(), h->row_start.end()), ed(d.E);
  for (int e = 0; e < d.E; e++) {
    ed[e] = (int)col[e] * (int)ls | ((int)shift[e] << 16);
  }
  PHY_HIP_CHECK(hipMalloc(&h->d_row_start, (d.M + 1) * sizeof(int)), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipMalloc(&h->d_edges, d.E * sizeof(int)), SRSRAN_ERROR);
  PHY_HIP_CHECK(upload(h->d_row_start, rs.data(), (d.M + 1) * sizeof(int)), SRSRAN_ERROR);
  PHY_HIP_CHECK(upload(h->d_edges, ed.data(), d.E * sizeof(int)), SRSRAN_ERROR);
  if (flood) {
    // edges of every variable node in row order (update_ldpc_soft_bits_c_flood walks the rows, ldpc_dec_c_flood.c:322-346)
    std::vector<int> cs(d.N + 1, 0), ce;
    for (int v = 0; v < d.N; v++) {
      cs[v] = (int)ce.size();
      for (int e = 0; e < d.E; e++) {
        if (col[e] == v) {
          ce.push_back(e | ((int)shift[e] << 16));
        }
      }
    }
    cs[d.N] = (int)ce.size();
    PHY_HIP_CHECK(hipMalloc(&h->d_col_start, cs.size() * sizeof(int)), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMalloc(&h->d_col_edges, ce.size() * sizeof(int)), SRSRAN_ERROR);
    PHY_HIP_CHECK(upload(h->d_col_start, cs.data(), cs.size() * sizeof(int)), SRSRAN_ERROR);
    PHY_HIP_CHECK(upload(h->d_col_edges, ce.data(), ce.size() * sizeof(int)), SRSRAN_ERROR);
  }
  {
    // slabs of check-to-variable messages, one per workgroup slot and code word it holds (<= 256 / Z words)
    const size_t es  = dtype == ldpc::DT_F32 ? 4 : (dtype == ldpc::DT_I16 ? 2 : 1);
    const size_t cpb = (size_t)choose_cpb(ls, (size_t)d.N * ls * (dtype == ldpc::DT_F32 ? 4 : 2));
    size_t       slots = ((size_t)(max_nof_cw ? max_nof_cw : 1) + cpb - 1) / cpb;
    slots = slots < 2048 ? slots : 2048; // resident workgroups (256 CUs x at most 5 ... 8) with room to spare; words are handed out by a counter
    h->cpb             = (int)cpb;
    h->slots           = (int)slots;
    PHY_HIP_CHECK(hipMalloc(&h->d_c2v, slots * cpb * d.E * ls * es), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipMalloc(&h->d_work, sizeof(unsigned int)), SRSRAN_ERROR);
  }
  *hh = h;
  return SRSRAN_SUCCESS;
}

extern "C" void srsran_hip_ldpc_batch_free(srsran_hip_ldpc_batch_t* h)
{
  if (!h) {
    return;
  }
  hipFree(h->d_row_start);
  hipFree(h->d_edges);
  hipFree(h->d_col_start);
  hipFree(h->d_col_edges);
  for (auto& kv : h->crc_mult) {
    hipFree(kv.second);
  }
  hipFree(h->d_c2v);
  hipFree(h->d_work);
  delete h;
}

static int ldpc_batch_run(srsran_hip_ldpc_batch_t* h, const void* d_llrs, uint32_t llr_stride, uint8_t* d_message,
                          uint32_t msg_stride, uint32_t n_cw, uint32_t cdwd_rm_length, uint8_t* d_iter_msgs, void* d_soft,
                          void* stream, uint32_t crc_poly = 0, int crc_order = 0, int* d_n_iter = nullptr, const uint32_t* d_cw_map = nullptr);

extern "C" int srsran_hip_ldpc_batch_run(srsran_hip_ldpc_batch_t* h, const int8_t* d_llrs, uint32_t llr_stride,
                                         uint8_t* d_message, uint32_t msg_stride, uint32_t n_cw,
                                         uint32_t cdwd_rm_length, uint8_t* d_iter_msgs, void* stream)
{
  if (h && h->dtype != ldpc::DT_I8) {
    set_error("ldpc batch: this object decodes %s LLRs, use srsran_hip_ldpc_batch_run_typed", h->dtype == ldpc::DT_F32 ? "float" : "int16");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  return ldpc_batch_run(h, d_llrs, llr_stride, d_message, msg_stride, n_cw, cdwd_rm_length, d_iter_msgs, nullptr, stream);
}

extern "C" int srsran_hip_ldpc_batch_run_typed(srsran_hip_ldpc_batch_t* h, const void* d_llrs, uint32_t llr_stride,
                                               uint8_t* d_message, uint32_t msg_stride, uint32_t n_cw,
                                               uint32_t cdwd_rm_length, uint8_t* d_iter_msgs, void* stream)
{
  return ldpc_batch_run(h, d_llrs, llr_stride, d_message, msg_stride, n_cw, cdwd_rm_length, d_iter_msgs, nullptr, stream);
}

// debug/parity entry point (tests bind it by name): also returns the a-posteriori soft bits (n_cw x bgN*Z elements)
extern "C" SRSRAN_API int srsran_hip_ldpc_batch_run_dbg(srsran_hip_ldpc_batch_t* h, const void* d_llrs, uint32_t llr_stride,
                                                        uint8_t* d_message, uint32_t msg_stride, uint32_t n_cw,
                                                        uint32_t cdwd_rm_length, void* d_soft, void* stream)
{
  return ldpc_batch_run(h, d_llrs, llr_stride, d_message, msg_stride, n_cw, cdwd_rm_length, nullptr, d_soft, stream);
}

static int ldpc_batch_run(srsran_hip_ldpc_batch_t* h, const void* d_llrs, uint32_t llr_stride, uint8_t* d_message,
                          uint32_t msg_stride, uint32_t n_cw, uint32_t cdwd_rm_length, uint8_t* d_iter_msgs, void* d_soft,
                          void* stream, uint32_t crc_poly, int crc_order, int* d_n_iter, const uint32_t* d_cw_map)
{
  TraceRange trace_("srsran_hip_ldpc_batch_run");
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_ldpc_batch_run", SRSRAN_ERROR);
  }
  if (h && n_cw == 0) {
    return SRSRAN_SUCCESS; // an empty batch is a no-op
  }
  if (!h || !d_llrs || !d_message || n_cw > (h->max_cw ? h->max_cw : 1)) {
    set_error("ldpc batch: invalid arguments");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  const uint32_t Z = h->Z, liftN = h->N * Z;
  // ldpc_decoder.c:51-65
  if (cdwd_rm_length > liftN - 2 * Z) {
    cdwd_rm_length = liftN - 2 * Z;
  }
  if (cdwd_rm_length < (uint32_t)(h->K + 2) * Z) {
    cdwd_rm_length = (h->K + 2) * Z;
  }
  if (cdwd_rm_length % Z) {
    cdwd_rm_length = (cdwd_rm_length / Z + 1) * Z;
  }
  const int n_layers = (uint8_t)(cdwd_rm_length / Z - h->K + 2); // :73
  if (n_cw > 1 && (llr_stride < liftN - 2 * Z || msg_stride < (uint32_t)h->K * Z)) {
    set_error("ldpc batch: strides too small");
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  ldpc::Params p;
  p.llrs       = d_llrs;
  p.msg        = d_message;
  p.iter_msgs  = d_iter_msgs;
  p.row_start  = h->d_row_start;
  p.edges      = h->d_edges;
  p.llr_stride = llr_stride;
  p.msg_stride = msg_stride;
  p.Z          = Z;
  p.bgN        = h->N;
  p.bgK        = h->K;
  p.n_layers   = n_layers;
  p.n_edges    = h->row_start[n_layers];
  p.max_iter   = h->max_iter;
  p.sf         = h->sf;
  p.n_cw       = (int)n_cw;
  p.dtype      = h->dtype;
  p.sf_f       = h->sf_f;
  p.c2v_ws     = h->d_c2v;
  p.work_counter = h->d_work; // code words are handed out by a counter (both kernels)
  PHY_HIP_CHECK(hipMemsetAsync(h->d_work, 0, sizeof(unsigned int), (hipStream_t)stream), SRSRAN_ERROR);
  p.soft_out   = d_soft;
  p.crc_poly   = 0;
  p.crc_order  = 0;
  p.crc_mult   = nullptr;
  p.n_iter_out = nullptr;
  p.cw_map     = d_cw_map;
  if (crc_order) {
    if (crc_order < 8 || crc_order > 24 || h->dtype != ldpc::DT_I8 || !d_n_iter) {
      set_error("ldpc batch: CRC early stop needs the int8 decoder, a generator of order 8..24 and an iteration array");
      return SRSRAN_ERROR_INVALID_INPUTS;
    }
    uint32_t*& d_mult = h->crc_mult[((uint64_t)crc_order << 32) | crc_poly];
    if (!d_mult) {
      // x^n mod g by repeated multiplication with x; lane c needs n = (Z - 1 - c) * bgK
      const uint32_t mask = (1u << crc_order) - 1u, g = crc_poly & mask;
      std::vector<uint32_t> m(Z);
      uint32_t              r = 1;
      for (int c = (int)Z - 1; c >= 0; c--) {
        m[c] = r;
        for (int k = 0; k < h->K; k++) {
          r = ((r << 1) & mask) ^ (((r >> (crc_order - 1)) & 1u) ? g : 0u);
        }
      }
      PHY_HIP_CHECK(hipMalloc(&d_mult, Z * sizeof(uint32_t)), SRSRAN_ERROR);
      // once per (object, generator); upload() returns when the device has the table (hip_common.h)
      PHY_HIP_CHECK(upload(d_mult, m.data(), Z * sizeof(uint32_t)), SRSRAN_ERROR);
    }
    p.crc_poly   = crc_poly;
    p.crc_order  = crc_order;
    p.crc_mult   = d_mult;
    p.n_iter_out = d_n_iter;
  }
  p.flood      = h->flood ? 1 : 0;
  p.n_col_edges = h->E;
  p.col_start  = h->d_col_start;
  p.col_edges  = h->d_col_edges;
  // with CRC early stop a workgroup lasts as long as its slowest code word: large lifting sizes then keep one word per
  // workgroup (the slab area holds cpb times as many single-word slabs)
  p.cpb        = (crc_order && Z > 128) ? 1 : h->cpb;
  p.max_slots  = h->slots * h->cpb / p.cpb;
  // (x * M) >> 9 == x * sf / 100 on 0..127 with x * M below 2^16: lets the packed kernel scale on the 16-bit multiplier
  p.sf_m9 = 0;
  {
    const int M = (h->sf * 512 + 99) / 100;
    bool      ok = M > 0 && 127 * M < 65536;
    for (int x = 0; ok && x <= 127; x++) {
      ok = ((x * M) >> 9) == x * h->sf / 100;
    }
    p.sf_m9 = ok ? M : 0;
  }
  p.packed  = 0;
  p.c2v_lds = 0;
  if (ldpc::packed_applies(p)) {
    // code words per workgroup for Z / 2 lanes per word (26 KB of soft words for BG1 Z = 384: four words, 12 waves)
    const int cap  = h->slots * h->cpb; // code-word slabs allocated
    // ... except when a word's Z / 2 lanes are whole waves (Z = 256, 384): one word per workgroup then -- five 3-wave workgroups per CU
    // drift out of phase and hide each other's message loads (Z = 384, 16,384 words, 20 iterations: 26.3 ms with four words per
    // workgroup, 25.6 ms with one; Z = 256: the same either way)
    int       pcpb = ((crc_order && Z > 128) || (Z >= 256 && (Z / 2) % 64 == 0)) ? 1 : choose_cpb((int)Z / 2, (size_t)h->N * (Z / 2) * 2);
    pcpb           = pcpb > cap ? cap : pcpb;
    if (const int v = knob(KNOB_LDPC_PCPB); v > 0) { // development knob
      pcpb        = (v > 0 && v <= cap && v * (int)(Z / 2) <= 768 && (size_t)v * h->N * Z <= 150 * 1024) ? v : pcpb;
    }
    p.cpb       = pcpb;
    p.max_slots = cap / pcpb;
    p.packed    = 1;
    // Small batches (a transport block's code blocks, a single srsran_ldpc_decoder_decode_c call): no more words than the chip has CUs, so a word may
    // have a CU's LDS to itself -- its check-to-variable messages then live there (BG1 Z = 384: 121 KB next to 26 KB of soft words) and a row waits
    // for LDS instead of the L2 / HBM round trip of its message loads (SRSRAN_HIP_LDPC_C2V_LDS=0: never)
    if ((int)n_cw <= 256 && knob(KNOB_LDPC_C2V_LDS) != 0 && ldpc::packed_c2v_lds_fits(p)) {
      p.cpb       = 1;
      p.max_slots = cap;
      p.c2v_lds   = 1;
    }
  }
  PHY_HIP_CHECK(ldpc::launch(p, (hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_ldpc_batch_run_crc(srsran_hip_ldpc_batch_t* h, const int8_t* d_llrs, uint32_t llr_stride, uint8_t* d_message,
                                             uint32_t msg_stride, uint32_t n_cw, uint32_t cdwd_rm_length, uint32_t crc_polynom, uint32_t crc_order,
                                             int32_t* d_nof_iterations, void* stream)
{
  if (!crc_order || !d_nof_iterations) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  return ldpc_batch_run(h, d_llrs, llr_stride, d_message, msg_stride, n_cw, cdwd_rm_length, nullptr, nullptr, stream, crc_polynom, (int)crc_order,
                        (int*)d_nof_iterations);
}

// the same with an indirection: code word i lives in row d_cw_map[i] of the LLR / message arrays (the NR transport-block loop
// decodes the undecoded code blocks of a soft buffer where they lie); d_nof_iterations stays indexed by i
extern "C" int srsran_hip_ldpc_batch_run_crc_map(srsran_hip_ldpc_batch_t* h, const int8_t* d_llrs, uint32_t llr_stride, uint8_t* d_message,
                                                 uint32_t msg_stride, const uint32_t* d_cw_map, uint32_t n_cw, uint32_t cdwd_rm_length,
                                                 uint32_t crc_polynom, uint32_t crc_order, int32_t* d_nof_iterations, void* stream)
{
  if (h) {
    PHY_DEV_GUARD(h->tag, "srsran_hip_ldpc_batch_run_crc_map", SRSRAN_ERROR);
  }
  if (!crc_order || !d_nof_iterations || !d_cw_map) {
    return SRSRAN_ERROR_INVALID_INPUTS;
  }
  return ldpc_batch_run(h, d_llrs, llr_stride, d_message, msg_stride, n_cw, cdwd_rm_length, nullptr, nullptr, stream, crc_polynom, (int)crc_order,
                        (int*)d_nof_iterations, d_cw_map);
}

// ------------------------------------------------------------------------------------------------ handle ABI

namespace {
struct LdpcCtx {
  DeviceTag tag;
  srsran_hip_ldpc_batch_t* b      = nullptr;
  hipStream_t              stream = nullptr;
  size_t                   esz    = 1;       // bytes per LLR (1 / 2 / 4)
  uint32_t                 n_iter = 0;       // iterations the device runs: max_nof_iter, twice that for the flooded schedule
  int8_t*                  d_llr  = nullptr;
  uint8_t*                 d_msg  = nullptr;
  uint8_t*                 d_iter = nullptr;
  int8_t*                  h_llr  = nullptr; // pinned
  uint8_t*                 h_msg  = nullptr; // pinned
  uint8_t*                 h_iter = nullptr; // pinned
  int                      dec_type = 0;     // srsran_ldpc_decoder_type_t given at init
  std::vector<uint8_t>     rec;              // one output record of the shared submission queue (message + iteration count)
};

uint32_t crc_bits(uint32_t poly, int order, const uint8_t* bits, int len)
{
  // crc.c:92-140 restated bitwise: zero init, MSB first, a byte counts as 1 when (int8) > 0 (crc.c:117)
  const uint64_t mask = ((((uint64_t)1 << (order - 1)) - 1) << 1) | 1;
  const uint64_t high = (uint64_t)1 << (order - 1);
  uint64_t       crc  = 0;
  for (int i = 0; i < len; i++) {
    uint64_t b   = ((int8_t)bits[i] > 0) ? 1 : 0;
    uint64_t top = (crc & high) ? 1 : 0;
    crc          = (crc << 1) & mask;
    if (top ^ b) {
      crc ^= (uint64_t)poly;
    }
  }
  return (uint32_t)(crc & mask);
}

void ldpc_ctx_free(void* o)
{
  auto*    q = reinterpret_cast<srsran_ldpc_decoder_t*>(o);
  LdpcCtx* c = reinterpret_cast<LdpcCtx*>(q->ptr);
  if (c) {
    srsran_hip_ldpc_batch_free(c->b);
    hipFree(c->d_llr);
    hipFree(c->d_msg);
    hipFree(c->d_iter);
    hipHostFree(c->h_llr);
    hipHostFree(c->h_msg);
    hipHostFree(c->h_iter);
    if (c->stream) {
      hipStreamDestroy(c->stream);
    }
    delete c;
  }
  if (q->pcm) {
    free(q->pcm);
  }
  if (q->var_indices) {
    free(q->var_indices);
  }
}

int ldpc_decode_any(void* o, const void* llrs, uint8_t* message, uint32_t cdwd_rm_length, srsran_crc_t* crc);

int ldpc_decode_c(void* o, const int8_t* llrs, uint8_t* message, uint32_t cdwd_rm_length, srsran_crc_t* crc)
{
  return ldpc_decode_any(o, llrs, message, cdwd_rm_length, crc);
}
int ldpc_decode_s(void* o, const int16_t* llrs, uint8_t* message, uint32_t cdwd_rm_length, srsran_crc_t* crc)
{
  return ldpc_decode_any(o, llrs, message, cdwd_rm_length, crc);
}
int ldpc_decode_f(void* o, const float* llrs, uint8_t* message, uint32_t cdwd_rm_length, srsran_crc_t* crc)
{
  return ldpc_decode_any(o, llrs, message, cdwd_rm_length, crc);
}

// LDPC_DECODER_TEMPLATE (ldpc_decoder.c:44-104) for any of the three LLR types
int ldpc_decode_any(void* o, const void* llrs, uint8_t* message, uint32_t cdwd_rm_length, srsran_crc_t* crc)
{
  auto*    q = reinterpret_cast<srsran_ldpc_decoder_t*>(o);
  LdpcCtx* c = reinterpret_cast<LdpcCtx*>(q->ptr);
  if (!c) {
    return -1;
  }
  PHY_DEV_GUARD(c->tag, "srsran_ldpc_decoder_decode", -1);
  const uint32_t n_llr     = q->liftN - 2 * q->ls; // init_ldpc_dec_c reads all of them
  const uint32_t liftK     = q->liftK;
  // callers inside this function right now (any handle): with a handful of them the private streams overlap their one-workgroup
  // launches perfectly (3 threads: 23 Mbit/s against 10 through the single-lane queue), beyond that the driver serialises the
  // submissions and the queue wins (16 threads: 35 against 32; profiles/r02_bench_handle.json)
  static std::atomic<int> g_inflight{0};
  struct Inflight {
    int n;
    Inflight() : n(++g_inflight) {}
    ~Inflight() { --g_inflight; }
  } inflight;
  if (coalescing_enabled() && inflight.n > 4) {
    // decodes of the same shape that are in flight on different handles share one batch launch (coalesce.h); a record of the
    // output staging area is the message followed by the iteration count of the CRC early stop
    const uint32_t nit_off = (liftK + 3u) & ~3u;
    const uint32_t poly = crc ? (uint32_t)crc->polynom : 0u, order = crc ? (uint32_t)crc->order : 0u;
    char           key[160];
    // the engine depends on decoder type, base graph, lifting size, scaling and iteration budget; the rate-matched length and the CRC
    // change with every grant and are run-time parameters of a launch: they travel as the request's grouping tag
    snprintf(key, sizeof(key), "ldpc:d%d:t%d:bg%d:z%u:sf%a:it%u", current_device(), c->dec_type, (int)q->bg, (unsigned)q->ls, (double)q->scaling_fctr, q->max_nof_iter);
    const uint64_t tag = (uint64_t)cdwd_rm_length | ((uint64_t)order << 16) | ((uint64_t)poly << 24);
    const size_t esz  = c->esz;
    std::shared_ptr<Coalescer> co = coalescer_for(key, [&]() -> Coalescer* {
      const uint32_t                   cap = 32;
      const srsran_basegraph_t         bgq = q->bg;
      const uint16_t                   lsq = q->ls;
      const float                      sfq = q->scaling_fctr;
      const uint32_t                   itq = q->max_nof_iter;
      const srsran_ldpc_decoder_type_t tyq = (srsran_ldpc_decoder_type_t)c->dec_type;
      const uint32_t in_stride = (uint32_t)(Coalescer::stride_of(n_llr * esz) / esz), out_stride = (uint32_t)Coalescer::stride_of(nit_off + 4);
      return new Coalescer(n_llr * esz, nit_off + 4, cap, 1, [=](int) -> Coalescer::Engine {
        srsran_hip_ldpc_batch_t* b     = nullptr;
        int*                     d_nit = nullptr;
        if (srsran_hip_ldpc_batch_create_typed(&b, bgq, lsq, sfq, itq, cap, tyq) != SRSRAN_SUCCESS || hipMalloc(&d_nit, cap * sizeof(int)) != hipSuccess) {
          if (b) {
            srsran_hip_ldpc_batch_free(b);
          }
          return Coalescer::Engine();
        }
        auto run = [=](const void* d_in, void* d_out, uint32_t n, uint64_t t, hipStream_t st) -> int {
          const uint32_t cdwd_rm_length = (uint32_t)(t & 0xffffu), order = (uint32_t)((t >> 16) & 0xffu), poly = (uint32_t)(t >> 24);
          if (order) {
            if (srsran_hip_ldpc_batch_run_crc(b, static_cast<const int8_t*>(d_in), in_stride, static_cast<uint8_t*>(d_out), out_stride, n, cdwd_rm_length,
                                              poly, order, d_nit, st)) {
              return SRSRAN_ERROR;
            }
            PHY_HIP_CHECK(hipMemcpy2DAsync(static_cast<uint8_t*>(d_out) + nit_off, out_stride, d_nit, sizeof(int), sizeof(int), n, hipMemcpyDeviceToDevice, st),
                          SRSRAN_ERROR);
            return SRSRAN_SUCCESS;
          }
          return srsran_hip_ldpc_batch_run_typed(b, d_in, in_stride, static_cast<uint8_t*>(d_out), out_stride, n, cdwd_rm_length, nullptr, st);
        };
        return Coalescer::Engine{run, [=]() {
                                   srsran_hip_ldpc_batch_free(b);
                                   (void)hipFree(d_nit);
                                 }};
      });
    });
    if (co) {
      std::vector<uint8_t>& rec = c->rec;
      rec.resize(nit_off + 4);
      if (co->submit(llrs, rec.data(), tag) != SRSRAN_SUCCESS) {
        fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_decoder: %s\n", get_error());
        return -1;
      }
      memcpy(message, rec.data(), liftK);
      int nit = (int)q->max_nof_iter;
      if (crc) {
        memcpy(&nit, rec.data() + nit_off, sizeof(int));
      }
      return nit;
    }
  }
  memcpy(c->h_llr, llrs, n_llr * c->esz);
  PHY_HIP_CHECK(hipMemcpyAsync(c->d_llr, c->h_llr, n_llr * c->esz, hipMemcpyHostToDevice, c->stream), -1);
  if (crc) {
    // CRC early stop on the device (ldpc_decoder.c:87-99): the kernel leaves the iteration loop at the first match and
    // reports the iteration count (0: no match within the budget, the reference's return value as well)
    int* d_nit = reinterpret_cast<int*>(c->d_iter);
    if (srsran_hip_ldpc_batch_run_crc(c->b, reinterpret_cast<const int8_t*>(c->d_llr), n_llr, c->d_msg, liftK, 1, cdwd_rm_length,
                                      (uint32_t)crc->polynom, (uint32_t)crc->order, d_nit, c->stream)) {
      fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_decoder: %s\n", get_error());
      return -1;
    }
    PHY_HIP_CHECK(hipMemcpyAsync(c->h_msg, c->d_msg, liftK, hipMemcpyDeviceToHost, c->stream), -1);
    PHY_HIP_CHECK(hipMemcpyAsync(c->h_iter, d_nit, sizeof(int), hipMemcpyDeviceToHost, c->stream), -1);
    PHY_HIP_CHECK(hipStreamSynchronize(c->stream), -1);
    const int nit = *reinterpret_cast<const int*>(c->h_iter);
    // the reference extracts the message at every iteration it checks: after a failed run `message` holds the last one
    memcpy(message, c->h_msg, liftK);
    return nit;
  }
  if (srsran_hip_ldpc_batch_run_typed(c->b, c->d_llr, n_llr, c->d_msg, liftK, 1, cdwd_rm_length, nullptr, c->stream)) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_decoder: %s\n", get_error());
    return -1;
  }
  PHY_HIP_CHECK(hipMemcpyAsync(c->h_msg, c->d_msg, liftK, hipMemcpyDeviceToHost, c->stream), -1);
  PHY_HIP_CHECK(hipStreamSynchronize(c->stream), -1);
  memcpy(message, c->h_msg, liftK);
  return (int)q->max_nof_iter;
}
} // namespace

extern "C" int srsran_ldpc_decoder_init(srsran_ldpc_decoder_t* q, const srsran_ldpc_decoder_args_t* args)
{
  if (q == NULL || args == NULL) {
    return -1;
  }
  BgDims d;
  if (ls_index_of(args->ls) == VOID_LIFTSIZE) {
    fprintf(stderr, "Invalid lifting size %d\n", args->ls);
    return -1;
  }
  if (!bg_dims(args->bg, &d)) {
    fprintf(stderr, "Base Graph BG%d does not exist\n", args->bg + 1);
    return -1;
  }
  size_t esz   = 1;
  bool   flood = false;
  switch (args->type) {
    case SRSRAN_LDPC_DECODER_F:
      esz = 4;
      break;
    case SRSRAN_LDPC_DECODER_S:
      esz = 2;
      break;
    case SRSRAN_LDPC_DECODER_C:
    case SRSRAN_LDPC_DECODER_C_AVX2:
    case SRSRAN_LDPC_DECODER_C_AVX512:
      break; // one family: int8 layered min-sum, identical results in the reference
    case SRSRAN_LDPC_DECODER_C_FLOOD:
    case SRSRAN_LDPC_DECODER_C_AVX2_FLOOD:
    case SRSRAN_LDPC_DECODER_C_AVX512_FLOOD:
      flood = true; // runs 2 * max_nof_iter iterations (ldpc_decoder.c:136)
      break;
    default:
      return -1; // ldpc_decoder.c:644-646
  }
  memset(q, 0, sizeof(*q));
  q->bg           = args->bg;
  q->bgN          = (uint8_t)d.N;
  q->bgM          = (uint8_t)d.M;
  q->bgK          = (uint8_t)d.K;
  q->ls           = args->ls;
  q->liftK        = (uint16_t)(args->ls * d.K);
  q->liftM        = (uint16_t)(args->ls * d.M);
  q->liftN        = (uint16_t)(args->ls * d.N);
  q->max_nof_iter = args->max_nof_iter == 0 ? 10 : args->max_nof_iter;
  q->pcm          = (uint16_t*)malloc(sizeof(uint16_t) * d.M * d.N);
  q->var_indices  = (int8_t(*)[MAX_CNCT])malloc(sizeof(int8_t[MAX_CNCT]) * d.M);
  if (!q->pcm || !q->var_indices || create_compact_pcm(q->pcm, q->var_indices, q->bg, q->ls)) {
    free(q->pcm);
    free(q->var_indices);
    memset(q, 0, sizeof(*q));
    return -1;
  }
  if (args->scaling_fctr <= 0 || args->scaling_fctr > 1) {
    perror("The scaling factor of the min-sum algorithm should be larger than 0 and not larger than 1.");
    free(q->pcm);
    free(q->var_indices);
    memset(q, 0, sizeof(*q));
    return -1;
  }
  q->scaling_fctr = args->scaling_fctr;

  auto* c = new LdpcCtx;
  q->ptr  = c;
  q->free = ldpc_ctx_free;
  c->esz  = esz;
  c->dec_type = (int)args->type;
  c->n_iter = (flood ? 2u : 1u) * q->max_nof_iter;
  // one decode entry point per object, as init_f / init_s / init_c register them (ldpc_decoder.c:170-260)
  if (esz == 4) {
    q->decode_f = ldpc_decode_f;
  } else if (esz == 2) {
    q->decode_s = ldpc_decode_s;
  } else {
    q->decode_c = ldpc_decode_c;
  }
  const uint32_t n_llr = q->liftN - 2 * q->ls, msg_bytes = (q->liftK + 7) / 8;
  bool ok = srsran_hip_ldpc_batch_create_typed(&c->b, q->bg, q->ls, q->scaling_fctr, q->max_nof_iter, 1, args->type) == SRSRAN_SUCCESS &&
            hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc(&c->d_llr, n_llr * esz) == hipSuccess && hipMalloc(&c->d_msg, q->liftK) == hipSuccess &&
            hipMalloc(&c->d_iter, (size_t)msg_bytes * c->n_iter) == hipSuccess &&
            hipHostMalloc(&c->h_llr, n_llr * esz) == hipSuccess && hipHostMalloc(&c->h_msg, q->liftK) == hipSuccess &&
            hipHostMalloc(&c->h_iter, (size_t)msg_bytes * c->n_iter) == hipSuccess;
  if (!ok) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_decoder_init: %s\n", get_error());
    srsran_ldpc_decoder_free(q);
    return -1;
  }
  return 0;
}

extern "C" void srsran_ldpc_decoder_free(srsran_ldpc_decoder_t* q)
{
  if (q->free) {
    q->free(q);
  }
  memset(q, 0, sizeof(srsran_ldpc_decoder_t));
}

extern "C" int srsran_ldpc_decoder_decode_f(srsran_ldpc_decoder_t* q, const float* llrs, uint8_t* message, uint32_t cdwd_rm_length)
{
  if (!q->decode_f) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_decoder_decode_f: the object was not initialised as SRSRAN_LDPC_DECODER_F\n");
    return -1;
  }
  return q->decode_f(q, llrs, message, cdwd_rm_length, NULL);
}

extern "C" int srsran_ldpc_decoder_decode_s(srsran_ldpc_decoder_t* q, const int16_t* llrs, uint8_t* message, uint32_t cdwd_rm_length)
{
  if (!q->decode_s) {
    fprintf(stderr, "[srsran_phy_hip] srsran_ldpc_decoder_decode_s: the object was not initialised as SRSRAN_LDPC_DECODER_S\n");
    return -1;
  }
  return q->decode_s(q, llrs, message, cdwd_rm_length, NULL);
}

extern "C" int srsran_ldpc_decoder_decode_c(srsran_ldpc_decoder_t* q, const int8_t* llrs, uint8_t* message, uint32_t cdwd_rm_length)
{
  if (!q->decode_c) {
    return -1;
  }
  return q->decode_c(q, llrs, message, cdwd_rm_length, NULL);
}

extern "C" int srsran_ldpc_decoder_decode_crc_c(srsran_ldpc_decoder_t* q, const int8_t* llrs, uint8_t* message,
                                                uint32_t cdwd_rm_length, srsran_crc_t* crc)
{
  if (!q->decode_c) {
    return -1;
  }
  return q->decode_c(q, llrs, message, cdwd_rm_length, crc);
}
