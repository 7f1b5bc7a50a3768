// sync_device.h -- kernel parameter blocks and launchers of sync_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace sync {

struct PssResult {
  int32_t peak_pos;   // index of the correlation maximum (srsran_pss_find_pss return value)
  float   peak_value; // conv_output_avg[peak]
  float   psr;        // peak / side-lobe ratio (pss.c:408-437)
};

struct SssResult {
  int32_t  available; // 0: the SSS symbol does not fit before the PSS peak
  uint32_t m0, m1;
  float    m0_value, m1_value;
  int32_t  N_id_1; // -1 if not found
  uint32_t sf_idx; // 0 or 5
};

struct PssParams {
  const void* in;      // n_cap captures of frame_size cf, in_stride cf apart
  const void* twiddle; // 4096 x cf, then 16 x 64 cf: the per-lane tables of pss_wave_kernel
  const void* filt;    // 3 x 4096 cf: DFT_4096 of the zero-padded time replica of each N_id_2, / 4096
  float*      corr;    // n_cap x 3 x corr_stride floats: |conv|^2 (moving average), zero beyond n_out
  float*      part_val; // n_cap x 3 x n_blocks
  int*        part_idx;
  void*       spec;     // n_cap x n_blocks x 4096 cf: spectrum of every block between its hypotheses (pss_wave_kernel)
  size_t      in_stride;
  size_t      corr_stride;
  int         n_cap;
  int         n_blocks;
  int         hop;        // outputs per block = 4096 - fft_size
  int         part_span;  // outputs behind one part_val / part_idx entry (hop, or 256 on the direct path)
  int         fft_size;
  int         frame_size;
  int         n_out;      // frame_size + fft_size - 2 (pss.c:493)
  int         n_id_2_mask;
  float       ema_alpha;
};

struct SssParams {
  const void*  in;
  const void*  twiddle; // fft_size x cf
  const float* s_tilde; // 31
  const float* c_tilde;
  const float* z_tilde;
  const int*   sss_pos; // optional n_cap x 3 explicit symbol positions; nullptr: derive from the PSS peak
  const void*  ce;      // optional n_cap x 3 x 62 cf channel estimates
  size_t       in_stride;
  int          n_cap;
  int          fft_size;
  int          frame_size;
  int          cp_len;     // CP of the symbols after the first (normal) or extended CP
  int          cp_ext_len; // extended CP length (space check of sync.c:735)
  int          M;          // 0 differential, 1 full, 3 partial
  int          n_id_2_mask;
  float        threshold;
};

struct CellResult { // == srsran_hip_cell_t (phy_sync_abi.h)
  int32_t  peak_pos;
  float    peak_value;
  float    psr;
  int32_t  sss_available;
  uint32_t m0, m1;
  float    m0_value, m1_value;
  int32_t  N_id_1;
  uint32_t sf_idx;
};

// ---- helpers of the srsran_sync_t glue
enum { DOT_PLAIN = 0, DOT_CONJ = 1, DOT_POWER = 2 }; // sum a*b | sum a*conj(b) | mean |a|^2
struct DotJobs {
  const void* a[8];
  const void* b[8];
  int         n[8];
  int         mode[8];
  int         count;
};
hipError_t launch_cmul(const void* a, const void* b, void* out, int n, bool conj_b, hipStream_t stream);
hipError_t launch_lincomb(const void* a, float sa, const void* b, float sb, void* out, int n, hipStream_t stream);
hipError_t launch_dots(const DotJobs& jobs, void* d_out, hipStream_t stream); // d_out: count x cf
hipError_t launch_cp_synch(const void* in, void* corr, int max_offset, int nof_symbols, int cp_len, int N, int* d_argmax, hipStream_t stream);
struct CfoSeeds { // start phases of the oscillators of srsran_vec_apply_cfo, evaluated on the host with cexpf
  float phase_re[8], phase_im[8];
  float osc8_re, osc8_im;
  float tail_re, tail_im;
  float osc1_re, osc1_im;
};
hipError_t launch_apply_cfo(const void* x, void* z, int len, const CfoSeeds& seeds, hipStream_t stream);
hipError_t launch_decim(const void* in, void* out, int n_out, int M, const float* taps4, hipStream_t stream);

hipError_t launch_pack(const PssResult* a, const SssResult* b, CellResult* out, int n, hipStream_t stream);
hipError_t launch_pss(const PssParams& p, PssResult* d_res, hipStream_t stream);
hipError_t launch_pss_wave_blocks(const PssParams& p, hipStream_t stream); // pss_wave_kernels.hip: corr / part_val / part_idx of every block
hipError_t launch_sss(const SssParams& p, const PssResult* d_pss, SssResult* d_res, hipStream_t stream);
hipError_t launch_pss_direct(const PssParams& p, const void* d_replica, PssResult* d_res, hipStream_t stream);

} // namespace sync
} // namespace phyhip
