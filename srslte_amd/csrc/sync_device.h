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
  const void* twiddle; // 4096 x cf
  const void* filt;    // 3 x 4096 cf: DFT_4096 of the zero-padded time replica of each N_id_2, / 4096
  float*      corr;    // n_cap x 3 x corr_stride floats: |conv|^2 (moving average), zero beyond n_out
  float*      part_val; // n_cap x 3 x n_blocks
  int*        part_idx;
  size_t      in_stride;
  size_t      corr_stride;
  int         n_cap;
  int         n_blocks;
  int         hop;        // outputs per block = 4096 - fft_size
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

hipError_t launch_pack(const PssResult* a, const SssResult* b, CellResult* out, int n, hipStream_t stream);
hipError_t launch_pss(const PssParams& p, PssResult* d_res, hipStream_t stream);
hipError_t launch_sss(const SssParams& p, const PssResult* d_pss, SssResult* d_res, hipStream_t stream);

} // namespace sync
} // namespace phyhip
