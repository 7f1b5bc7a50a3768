// modem_device.h -- parameter blocks and launcher of modem_kernels.hip (soft demodulation + Gold-sequence descrambling)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace modem {

enum { LLR_I16 = 0, LLR_I8 = 1, LLR_F32 = 2 };
enum { MOD_PASS = 5 }; // no demodulation: the input already holds LLRs of the output type (srsran_sequence_apply_*)

// Gold sequence tables.  x1 does not depend on the seed: its chips (after Nc = 1600) are stored packed.  The x2 register
// at chip 1600 + 128 j is linear in the 31 seed bits: one column per bit (sequence.c:150-181 keeps the same thing for
// j = 0 only).  2 MB + 256 KB, resident in L2 / Infinity Cache.
#define MODEM_SEQ_CHUNK 128u
#define MODEM_SEQ_NCHUNKS 16384u // 2^21 chips per sequence
#define MODEM_TILE_SYMS 2048u    // symbols per workgroup
#define MODEM_TILE_BITS (8u * MODEM_TILE_SYMS) // soft bits per workgroup (256-QAM: 8 per symbol)

struct Job {
  uint32_t mod;     // srsran_mod_t or MOD_PASS
  uint32_t n;       // symbols (soft bits for MOD_PASS)
  uint32_t in_off;  // first symbol (cf_t units) / first input soft bit
  uint32_t out_off; // first output soft bit
  uint32_t seed;    // c_init
  uint32_t scramble;
  uint32_t tile0; // first workgroup of this job
  uint32_t ntiles;
  // UL-SCH channel de-interleaver (TS 36.212 5.2.2.8; sch.c:995-1018 ulsch_deinterleave without RI bits) folded into the store: the job's symbols
  // are a matrix of il_cols SC-FDMA symbols x il_rows sub-carriers read column by column (symbol s = column s / il_rows, row s % il_rows);
  // the soft bits of symbol s go to symbol slot (s % il_rows) * il_cols + s / il_rows (row by row).  0 = none.
  uint32_t il_rows;
  uint32_t il_cols;
};

struct Consts { // thresholds of demod_soft.c, evaluated by the host compiler exactly as the reference's are
  float t16_tail_s, t16_tail_b; // 2*SCALE/sqrtf(10) (float, 16-QAM scalar tails)
  float f16, f64a, f64b;        // 2/sqrtf(10), 4/sqrtf(42), 2/sqrtf(42)
  float c8, c4, c2;             // 8,4,2 / sqrtf(170)
  float qpsk_s, qpsk_b, qpsk_f; // -SCALE*M_SQRT2 as float
  int   o16_s, o64a_s, o64b_s, o16_b, o64a_b, o64b_b;
};

struct Params {
  const void*     in;
  void*           out;
  const Job*      jobs; // device array, or nullptr: `single`
  Job             single;
  uint32_t        n_jobs;
  uint32_t        n_tiles;
  int             llr_type;
  const uint32_t* tile_job; // job index of every workgroup (device array; unused with `single`)
  const uint32_t* x1_bits;  // MODEM_SEQ_NCHUNKS x 4 words
  const uint32_t* x2_cols;  // MODEM_SEQ_NCHUNKS x 31
  Consts          k;
};

// ---- modulator: packed bits (MSB first) -> scrambling with c_init = seed (optional) -> constellation point (x scale), what
// pdsch.c:1005-1018 chains as srsran_sequence_pdsch_apply_pack + srsran_mod_modulate_bytes (+ srsran_vec_sc_prod_cfc, :1119)
struct ModParams {
  const uint8_t*  bits;    // nof_symbols * Qm bits, byte packed, MSB first
  float2*         out;     // nof_symbols constellation points
  const float2*   table;   // constellation tables of all modulations (mod_table_offset)
  uint32_t        mod;     // srsran_mod_t
  uint32_t        n;       // symbols
  uint32_t        seed;    // c_init
  uint32_t        scramble;
  float           scale;   // 1.0f: none
  const uint32_t* x1_bits;
  const uint32_t* x2_cols;
  // several code words in one launch (the grants of a TTI, chan_host.cpp): jobs != nullptr replaces mod / n / seed / scramble / scale per job, `bits` and
  // `out` are the bases the jobs' offsets count from; tile_job[workgroup] = its job (device-readable arrays)
  const struct ModJob* jobs;
  const uint32_t*      tile_job;
};
struct ModJob {
  uint32_t mod, n, seed, scramble;
  float    scale;
  uint32_t bits_off; // first byte of the job's packed bits
  uint32_t out_off;  // first constellation point
  uint32_t tile0;    // first workgroup of the job
};
__host__ __device__ inline uint32_t mod_table_offset(uint32_t mod) // BPSK 2 | QPSK 4 | 16-QAM 16 | 64-QAM 64 | 256-QAM 256 points
{
  return mod == 0 ? 0u : (mod == 1 ? 2u : (mod == 2 ? 6u : (mod == 3 ? 22u : 86u)));
}
hipError_t launch_mod(const ModParams& p, hipStream_t stream);
hipError_t launch_mod_jobs(const ModParams& p, uint32_t n_tiles, hipStream_t stream); // p.jobs / p.tile_job set
// srsran_sequence_apply_packed (sequence.c): out = in ^ c, byte-packed bits, MSB first; nbits is rounded up to whole 32-bit words (both buffers must hold them)
hipError_t launch_scramble_packed(const uint8_t* in, uint8_t* out, uint32_t nbits, uint32_t seed, const uint32_t* x1_bits, const uint32_t* x2_cols, hipStream_t stream);
// host side (modem_host.cpp): parameter block with the sequence tables and thresholds filled in; the constellation tables on the device
bool          params_for(Params& p, int llr_type);
const float2* mod_tables();
void          host_mod_table(uint32_t mod, float2* out);

uint32_t   tiles_of(uint32_t mod, uint32_t n);
hipError_t launch(const Params& p, hipStream_t stream);
// srsran_predecoding_single on device buffers (y, h, x: cf_t, 16-byte aligned; csi optional)
hipError_t launch_eq(const void* y, const void* h, void* x, float* csi, uint32_t n, float scaling, float noise, hipStream_t stream);

// the same for several grants at once: job j covers the symbol pairs [jobs[j - 1].end_pair, jobs[j].end_pair) of the three arrays with its own noise
// estimate (add_noise as launch_eq derives it: noise > 0); `jobs` must be readable by the device (device memory or a pinned image)
struct EqJob {
  uint32_t end_pair;
  float    noise;
  uint32_t add_noise;
};
hipError_t launch_eq_jobs(const void* y, const void* h, void* x, const EqJob* jobs, uint32_t n_jobs, uint32_t n_pairs, float scaling, hipStream_t stream);

} // namespace modem
} // namespace phyhip
