// ofdm_device.h -- kernel parameter block and launchers of ofdm_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace ofdm {

struct Params {
  const void* in;      // rx: n_sf x sf_sz time samples (cf);  tx: n_sf x nsym_sf x nof_re REs
  void*       out;     // rx: n_sf x nsym_sf x nof_re REs;     tx: n_sf x sf_sz time samples
  const void* twiddle; // N x cf: e^{-j 2 pi i / N}
  const void* shift;   // sf_sz x cf frequency-shift table (ofdm.c:334-356) or nullptr
  const void* ramp;    // N x cf window-offset ramp (rx only) or nullptr
  long        n_sym_total; // n_sf * nsym_sf
  int         N;
  int         nsym_sf; // symbols per subframe (14 / 12)
  int         slot_sz;
  int         sf_sz;
  int         cp0; // CP of the first symbol of a slot
  int         cp1; // CP of the others
  int         nof_re;
  int         dc;    // 1: bin 0 is the unused DC carrier
  int         win_n; // rx: FFT window advanced by this many samples into the CP
  int         spw;   // symbols per workgroup (set by the launcher)
  float       norm;  // 1/sqrt(N), or 0 for no normalisation
  // MBSFN subframe (ofdm.c:424-437,538-555): slot 0 holds 6 symbols at these positions / CP lengths, is
  // processed without window offset; slot 1 is a regular extended-CP slot
  int         mbsfn;
  int         mpos[6];
  int         mcp[6];
};

bool       size_supported(int n);
hipError_t launch(const Params& p, bool tx, hipStream_t stream);
// out[i] = a[i] * b[i] (complex), used by the handle API to reproduce the in-place shift of in_buffer
hipError_t launch_prod_ccc(const void* a, const void* b, void* out, int n, hipStream_t stream);

} // namespace ofdm
} // namespace phyhip
