// dft_device.h -- kernel parameter block and launcher of dft_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace dft {

struct Params {
  const void* in;
  void*       out;
  const void* twiddle; // N x cf: e^{-j 2 pi i / N}
  long        idist, odist; // distance between transforms (in cf)
  int         istride, ostride;
  int         how_many;
  int         N;
  int         npass;     // 0: N is not 2^a 3^b 5^c -> direct evaluation
  int         radix[16];
  int         backward;
  int         mirror, dc, db;
  float       norm; // 1/sqrt(N) or 0 (real transforms: 1/N, dft_fftw.c:375)
  int         real_mode; // 0 complex; 1 real -> half-complex (FFTW_R2HC); 2 half-complex -> real (FFTW_HC2R)
};

hipError_t launch(const Params& p, hipStream_t stream); // picks a compile-time plan (dft_fixed_kernels.hip) when one applies
bool       has_fixed_plan(const Params& p);
hipError_t launch_fixed(const Params& p, hipStream_t stream);
hipError_t launch_large_twiddle(void* y, int N1, int N2, bool backward, hipStream_t stream);
hipError_t launch_large_reorder(const void* in, void* out, int N, bool backward, bool dc, hipStream_t stream);

} // namespace dft
} // namespace phyhip
