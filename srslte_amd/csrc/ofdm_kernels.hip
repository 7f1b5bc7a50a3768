// ofdm_kernels.hip -- batched OFDM demodulator / modulator (srsran_ofdm_rx_sf / srsran_ofdm_tx_sf) for gfx950.
//
// Reference behaviour: lib/src/phy/dft/ofdm.c:387-422,453-466 (rx), :487-536,562-576 (tx).
// One kernel per direction does everything the reference does in 3-4 passes over memory:
//   rx: [freq-shift multiply] -> CP removal -> N-point FFT -> [window-offset phase ramp] -> fftshift +
//       DC removal + guard removal -> [1/sqrt(N)]            (HBM: reads 15N samples once... the CP is
//       never fetched; writes nof_re REs per symbol)
//   tx: RE scatter around DC (zero guards) -> IFFT -> [1/sqrt(N)] -> CP insertion -> [freq-shift]
// HBM traffic is the algorithmic minimum; everything else stays in registers/LDS (fft_device.h).
#include "fft_device.h"
#include "hip_common.h"
#include "ofdm_device.h"

namespace phyhip {
namespace ofdm {

using namespace fft;

// time samples are read exactly once: non-temporal loads keep them out of L2 (N = 2048: 0.369 -> 0.360 ms per 5040 subframes, A/B on one
// box with two builds selected through SRSRAN_HIP_LIB; non-temporal stores of the resource elements made no difference)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 stream_load(const float2* p)
{
  const f2v v = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
  return make_float2(v.x, v.y);
}

struct RxLoad {
  const float2* in;    // first sample of the FFT window of this symbol
  const float2* shift; // shift table at the same subframe position, or nullptr
  __device__ __forceinline__ float2 operator()(int n) const
  {
    float2 x = stream_load(in + n);
    if (shift) {
      x = cmul(x, shift[n]); // srsran_vec_prod_ccc(in, shift_buffer) ofdm.c:455-457
    }
    return x;
  }
};

struct RxStore {
  float2*       out;  // nof_re REs of this symbol
  const float2* ramp; // window-offset ramp e^{+j 2 pi n_win f / N} or nullptr (ofdm.c:130-138,405-407)
  float         norm; // 1/sqrt(N) or 0 = no normalisation
  int           N, half_re, dc;
  __device__ __forceinline__ void operator()(int f, float2 v) const
  {
    int k;
    if (f >= N - half_re) {
      k = f - (N - half_re); // ofdm.c:410
    } else if (f >= dc && f < dc + half_re) {
      k = half_re + f - dc; // ofdm.c:411
    } else {
      return;
    }
    if (ramp) {
      v = cmul(v, ramp[f]);
    }
    if (norm != 0.0f) {
      v = cscale(v, norm); // ofdm.c:414-416
    }
    out[k] = v;
  }
};

struct TxLoad {
  const float2* in; // nof_re REs of this symbol
  int           N, half_re, dc;
  __device__ __forceinline__ float2 operator()(int f) const
  {
    if (f >= N - half_re) {
      return in[f - (N - half_re)]; // ofdm.c:516
    } else if (f >= dc && f < dc + half_re) {
      return in[half_re + f - dc]; // ofdm.c:515
    }
    return make_float2(0.f, 0.f);
  }
};

struct TxStore {
  float2*       out;   // first sample of the useful part of this symbol
  const float2* shift; // shift table at the same position or nullptr (ofdm.c:573-575)
  float         norm;
  int           N, cp;
  __device__ __forceinline__ void operator()(int n, float2 v) const
  {
    if (norm != 0.0f) {
      v = cscale(v, norm); // ofdm.c:527-529
    }
    out[n] = shift ? cmul(v, shift[n]) : v;
    if (n >= N - cp) { // cyclic prefix, ofdm.c:532
      out[n - N] = shift ? cmul(v, shift[n - N]) : v;
    }
  }
};

template <class P, bool TX>
__global__ __launch_bounds__(256) void ofdm_kernel(const Params p)
{
  extern __shared__ float2 lds_all[];
  constexpr int N = P::N, T = P::T;
  const int     sym_local = threadIdx.x / T;
  const int     tid       = threadIdx.x - sym_local * T;
  const long    sym       = (long)blockIdx.x * p.spw + sym_local;
  const bool    active    = sym_local < p.spw && sym < p.n_sym_total;
  float2*       lds       = lds_all + sym_local * lds_elems(N);

  // symbol -> (subframe, slot, symbol in slot); useful part starts at cp0 + l*(N+cp1) inside the slot
  const long sf  = active ? sym / p.nsym_sf : 0;
  const int  l   = active ? (int)(sym - sf * p.nsym_sf) : 0;
  const int  half = p.nsym_sf >> 1;
  const int  slot = l / half, li = l - slot * half;
  int        pos  = slot * p.slot_sz + p.cp0 + li * (N + p.cp1); // sample index inside the subframe
  int        cp   = li == 0 ? p.cp0 : p.cp1;
  int        win  = p.win_n;
  const float2* in   = reinterpret_cast<const float2*>(p.in);
  float2*       out  = reinterpret_cast<float2*>(p.out);
  const float2* tw   = reinterpret_cast<const float2*>(p.twiddle);
  const float2* sh   = reinterpret_cast<const float2*>(p.shift);
  const float2* ramp = reinterpret_cast<const float2*>(p.ramp);
  if (p.mbsfn && slot == 0) {
    pos  = p.mpos[li];
    cp   = p.mcp[li];
    win  = 0;
    ramp = nullptr;
  }

  if (!TX) {
    const int wpos = pos - win;
    RxLoad    ld{in + sf * p.sf_sz + wpos, sh ? sh + wpos : nullptr};
    RxStore   st{out + (sf * p.nsym_sf + l) * (long)p.nof_re, win ? ramp : nullptr, p.norm, N, p.nof_re >> 1, p.dc};
    transform<P, false>(lds, tid, active, tw, ld, st);
  } else {
    TxLoad  ld{in + (sf * p.nsym_sf + l) * (long)p.nof_re, N, p.nof_re >> 1, p.dc};
    TxStore st{out + sf * p.sf_sz + pos, sh ? sh + pos : nullptr, p.norm, N, cp};
    transform<P, true>(lds, tid, active, tw, ld, st);
  }
}

template <class P>
static hipError_t launch_plan(const Params& p, bool tx, hipStream_t stream)
{
  Params q = p;
  q.spw    = 256 / P::T > 0 ? 256 / P::T : 1;
  int threads = q.spw * P::T;
  threads     = ((threads + 63) / 64) * 64;
  const size_t lds = (size_t)q.spw * lds_elems(P::N) * sizeof(float2);
  dim3 grid((unsigned)((q.n_sym_total + q.spw - 1) / q.spw));
  if (tx) {
    hipLaunchKernelGGL((ofdm_kernel<P, true>), grid, dim3(threads), lds, stream, q);
  } else {
    hipLaunchKernelGGL((ofdm_kernel<P, false>), grid, dim3(threads), lds, stream, q);
  }
  return hipGetLastError();
}

__global__ void prod_ccc_kernel(const float2* a, const float2* b, float2* o, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    o[i] = cmul(a[i], b[i]);
  }
}

hipError_t launch_prod_ccc(const void* a, const void* b, void* out, int n, hipStream_t stream)
{
  hipLaunchKernelGGL(prod_ccc_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float2*)a, (const float2*)b,
                     (float2*)out, n);
  return hipGetLastError();
}

bool size_supported(int n)
{
  switch (n) {
    case 128:
    case 256:
    case 384:
    case 512:
    case 768:
    case 1024:
    case 1536:
    case 2048:
    case 3072:
    case 4096:
      return true;
    default:
      return false;
  }
}

hipError_t launch(const Params& p, bool tx, hipStream_t stream)
{
  switch (p.N) {
    case 128:
      return launch_plan<Plan<128, 8, 16, 8, 1, 1>>(p, tx, stream);
    case 256:
      return launch_plan<Plan<256, 16, 16, 16, 1, 1>>(p, tx, stream);
    case 384:
      return launch_plan<Plan<384, 24, 16, 8, 3, 1>>(p, tx, stream);
    case 512:
      return launch_plan<Plan<512, 64, 8, 8, 8, 1>>(p, tx, stream);
    case 768:
      return launch_plan<Plan<768, 48, 16, 16, 3, 1>>(p, tx, stream);
    case 1024:
      return launch_plan<Plan<1024, 64, 16, 8, 8, 1>>(p, tx, stream);
    case 1536:
      return launch_plan<Plan<1536, 96, 16, 8, 4, 3>>(p, tx, stream);
    case 2048:
      return launch_plan<Plan<2048, 128, 16, 16, 8, 1>>(p, tx, stream);
    case 3072:
      return launch_plan<Plan<3072, 192, 16, 16, 4, 3>>(p, tx, stream);
    case 4096:
      return launch_plan<Plan<4096, 256, 16, 16, 16, 1>>(p, tx, stream);
    default:
      return hipErrorInvalidValue;
  }
}

} // namespace ofdm
} // namespace phyhip
