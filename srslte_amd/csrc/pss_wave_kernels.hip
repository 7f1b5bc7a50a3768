// pss_wave_kernels.hip -- PSS correlation (pss.c:446-534, convolution.c:113-120), one WAVE per overlap-save block.
//
// Same arithmetic plan as pss_block_kernel (sync_kernels.hip): 4096-point overlap-save blocks, the block is transformed
// once, multiplied by the cached spectrum of each N_id_2 replica, transformed back, |.|^2 (+ the moving average of
// pss.c:496-503), block arg-max.  What differs is where the transform lives:
//   * 4096 = 64 x 64.  Lane l of ONE wave holds the 64 points n = l + 64 r in registers and runs a complete 64-point
//     FFT on them without talking to anybody (8 x 8, all indices and inner twiddles compile-time constants); one
//     multiplication by W_4096^(l k), ONE 64 x 64 transposition through LDS (wave-private: no barrier, 65-element rows:
//     no bank conflict), and a second register FFT gives X[l + 64 k] -- the layout the block was loaded in.  A 4096-point
//     transform therefore costs one LDS round trip instead of the two workgroup-wide exchanges + barriers of the
//     three-pass radix-16 engine, and nothing waits for another wave.
//   * complex numbers are two plain f32 registers; +-i is a renaming of registers, a complex multiplication two v_mul + two v_fma.
//     (The first version ran on the packed f32 pipe, one instruction per complex addition: slower, see `cx` below; the file is
//     compiled without the SLP vectoriser so that it stays unpacked, srslte_amd/build.py.)
//   * TWO waves per SIMD (235 registers): a wave issues at most one vector instruction per four cycles, the SIMD takes one per
//     two, and a lone wave stands still through every memory wait.  The spectrum of the block serves three hypotheses but does not
//     fit next to the transform's working set at that occupancy: it is parked in global memory in between (pss_wave_kernel).
//   * every global access is 512 contiguous bytes per wave instruction (capture, filter spectra, parked spectrum, correlation row);
//     block borders, the zero padding of the convolution and the ragged last block are the bounds checks of buffer instructions.
// A measured alternative with two waves per block and three waves per SIMD (pss_pair_kernel) is at the end of the file.
#include "fft_reg.h"
#include "hip_common.h"
#include "sync_device.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace phyhip {
namespace sync {

using namespace regfft;

// 4096-point transform of one wave: v[r] of lane l = element l + 64 r, in and out.  T1[b] = W^(l b), T2[a] = W^(8 l a),
// W = e^{-2 pi i / 4096}; `img` = the wave's 64 x 65 LDS image of floats: real and imaginary parts cross it one after the other
// (16.6 KB per wave, so that eight waves fit a CU; LDS operations of one wave execute in order, no barrier is involved).
template <bool INV>
static __device__ __forceinline__ void fft4096_first(cx (&v)[64], const cx (&T1)[8], const cx (&T2)[8], float* img, int lane) // ... up to the transposition
{
  fft64<INV>(v);
#pragma unroll
  for (int k = 0; k < 64; k++) {
    if (k != 0) {
      const cx w = (k & 7) == 0 ? T2[k >> 3] : ((k >> 3) == 0 ? T1[k & 7] : cmul2<false>(T2[k >> 3], T1[k & 7]));
      v[k]       = cmul2<INV>(v[k], w);
    }
    img[k * 65 + lane] = v[k].x;
    if ((k & 7) == 7) {
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int n = 0; n < 64; n++) {
    v[n].x = img[lane * 65 + n];
  }
  __builtin_amdgcn_sched_barrier(0); // (the LDS operations of a wave execute in issue order: every read above precedes every write below)
#pragma unroll
  for (int k = 0; k < 64; k++) {
    img[k * 65 + lane] = v[k].y;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < 64; n++) {
    v[n].y = img[lane * 65 + n];
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <bool INV>
static __device__ __forceinline__ void fft4096(cx (&v)[64], const cx (&T1)[8], const cx (&T2)[8], float* img, int lane)
{
  fft4096_first<INV>(v, T1, T2, img, lane);
  fft64<INV>(v);
}

// One wave per overlap-save block and TWO waves per SIMD: a lone wave issues one vector instruction every four cycles, half of what
// the SIMD takes, and stands still through every memory wait.  Two need <= 256 registers each, which leaves no room for the
// block's spectrum next to the transform's working set -- it waits in global memory (see below).
// RECOMP: the block's spectrum is not parked at all -- every hypothesis transforms the block again (it comes from L2, where this very wave has
// just put it): 1.5x the transform work per block against 96 KB less HBM traffic (park once + read back twice: 62 % of what the kernel moves).
template <bool RECOMP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void pss_wave_kernel(const PssParams p)
{
  __shared__ float img[64 * 65];
  const int        cap = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
  const cx*        x   = reinterpret_cast<const cx*>(p.in) + (size_t)cap * p.in_stride;
  const cx*        tw  = reinterpret_cast<const cx*>(p.twiddle);

  cx T1[8], T2[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { // (the host stores these behind the 4096 twiddles in [entry][lane] order)
    T1[j] = tw[4096 + 64 * j + lane];
    T2[j] = tw[4096 + 64 * (8 + j) + lane];
  }

  const int i0  = blk * p.hop;           // first convolution output of this block
  const int off = i0 - (p.fft_size - 1); // capture index of segment element 0
  // the capture as a raw buffer: elements before its start (negative offsets wrap to huge ones) and behind its end read as zero,
  // which is the zero padding of the convolution -- no compare, no branch
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<cx*>(x), 0, p.frame_size * 8, 0x00020000);

  const int m_lo = p.fft_size - 1, m_hi = p.fft_size - 1 + p.hop;
  bool      have_spec = false;
  for (int h = 0; h < 3; h++) {
    if (!(p.n_id_2_mask & (1 << h))) {
      continue;
    }
    // Nothing below may look loop-invariant: hoisted out of the hypothesis loop, the 64 twiddle products, the 64 output offsets and
    // the 64 row predicates are several hundred registers that stay live across the whole loop.  Passing the lane index, the twiddle
    // tables and the bounds through an empty asm makes them values of this iteration.
    int lane_h = lane, m_lo_h = m_lo;
    asm volatile("" : "+v"(lane_h), "+s"(m_lo_h));
#pragma unroll
    for (int j = 0; j < 8; j++) {
      asm volatile("" : "+v"(T1[j].x), "+v"(T1[j].y), "+v"(T2[j].x), "+v"(T2[j].y)); // (in place: no second copy of the tables)
    }
    cx        v[64];
    const cx* filt = reinterpret_cast<const cx*>(p.filt) + (size_t)h * 4096 + lane_h;
    cx        f[16];
    // The filter spectrum of a hypothesis comes in batches of 16 values.  (Fenced: short of registers, the compiler otherwise issues one
    // load, waits for it, uses it -- 64 L2 round trips in a row per hypothesis: 1.3 of 2.8 ms per 256 captures.)
    cx* spec = reinterpret_cast<cx*>(p.spec) + ((size_t)cap * p.n_blocks + blk) * 4096 + lane_h;
    if (!have_spec) {
#pragma unroll
      for (int r = 0; r < 64; r++) {
        v[r] = __builtin_bit_cast(cx, __builtin_amdgcn_raw_buffer_load_b64(xb, (off + lane_h + 64 * r) * 8, 0, 0));
      }
      fft4096_first<false>(v, T1, T2, img, lane_h);
      // the first batch is requested before the second half of the forward transform and arrives under it
#pragma unroll
      for (int j = 0; j < 16; j++) {
        f[j] = filt[64 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
      fft64<false>(v);
      __builtin_amdgcn_sched_barrier(0);
      // The spectrum of the block serves every hypothesis, but two waves per SIMD have no registers to keep it (and one wave per SIMD
      // issues at half rate): it is parked in the block's 32 KB of the spectrum area and comes back from L2 / the Infinity Cache in
      // place of the block itself -- the same 64 loads the forward transform would start with, and no second and third transform.
      // (Requesting it row by row while the previous hypothesis is post-processed, into the registers that become free, was built too:
      // the values carried around the loop cost 100 spilled registers and the kernel took 2.3 instead of 1.4 ms.)
      if (!RECOMP && (p.n_id_2_mask >> (h + 1)) != 0) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
          spec[64 * r] = v[r];
        }
      }
      have_spec = !RECOMP;
    } else {
#pragma unroll
      for (int r = 0; r < 64; r++) {
        v[r] = spec[64 * r];
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        f[j] = filt[64 * j];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        v[16 * g + j] = cmul2<false>(v[16 * g + j], f[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g < 3) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          f[j] = filt[64 * (16 * (g + 1) + j)];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (the inverse transform wants the same 64 twiddle products as the forward one: common-subexpression elimination would keep all
    // of them alive in between -- 128 registers; the tables go through the empty asm again instead)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      asm volatile("" : "+v"(T1[j].x), "+v"(T1[j].y), "+v"(T2[j].x), "+v"(T2[j].y));
    }
    fft4096<true>(v, T1, T2, img, lane_h);

    // the correlation row as a raw buffer of n_out floats: lanes outside the block's share get an offset beyond it, and the
    // hardware drops their stores (and returns zero for their loads)
    const __amdgpu_buffer_rsrc_t cb =
        __builtin_amdgcn_make_buffer_rsrc(p.corr + ((size_t)cap * 3 + h) * p.corr_stride, 0, p.n_out * 4, 0x00020000);
    float best  = -1.0f;
    int   besti = 0x7fffffff;
    auto  post  = [&](auto ema_tag) {
      constexpr bool EMA = decltype(ema_tag)::value;
#pragma unroll
      for (int r = 0; r < 64; r++) {
        if (64 * r + 63 < m_lo_h) { // wave-uniform: the whole register row lies in the discarded head of the block
          continue;
        }
        const int  m = lane_h + 64 * r, i = off + m;
        const bool ok = m >= m_lo_h && m < m_hi && i < p.n_out;
        const int  bo = ok ? i * 4 : -1;
        float      pw = v[r].x * v[r].x + v[r].y * v[r].y; // srsran_vec_abs_square_cf
        if (EMA) {
          const float old = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cb, bo, 0, 0));
          pw              = pw * p.ema_alpha + old * (1.0f - p.ema_alpha); // pss.c:497-500
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pw), cb, bo, 0, 0);
        const bool gt = ok && pw > best; // i grows with r: the first maximum of the lane stays
        best          = gt ? pw : best;
        besti         = gt ? i : besti;
      }
    };
    if (p.ema_alpha > 0.0f && p.ema_alpha < 1.0f) {
      post(std::true_type{});
    } else {
      post(std::false_type{});
    }
    // block arg-max (first maximum wins on ties, as srsran_vec_max_fi)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob  = __shfl_down(best, o);
      const int   obi = __shfl_down(besti, o);
      if (ob > best || (ob == best && obi < besti)) {
        best  = ob;
        besti = obi;
      }
    }
    if (lane == 0) {
      const size_t o = ((size_t)cap * 3 + h) * p.n_blocks + blk;
      p.part_val[o]  = best;
      p.part_idx[o]  = besti;
    }
  }
}

#ifdef SRSRAN_HIP_WITH_VARIANTS // measured-and-rejected alternatives are compiled into tools/probe/lib/libsrsran_phy_hip_variants.so only (srslte_amd/build.py --variants)
// ================================================================================================================================
// MEASURED ALTERNATIVE (SRSRAN_HIP_PSS_VARIANT=pair), not the product: two waves per block, 128 lanes x 32 points.  The one-wave
// kernel above cannot have more than two waves per SIMD (128 registers of data per lane) and a wave issues at most one vector
// instruction per four cycles where the SIMD takes one per two, so its SIMDs idle whenever one of the two waves waits.  Here a lane
// holds 32 points (64 registers), three waves fit a SIMD, and the 4096-point transform is 32 x 32 x 4 with two LDS transpositions
// across the two waves.  Result (256 captures of 10 ms, one box): 1.58 ms per step against 1.39 ms -- the eight workgroup barriers per
// transform (one 16.9 KB image, real and imaginary parts in turn; two images would cost the third wave) leave each wave active 39 % of
// its time instead of 70 %, which the third wave only just makes up for (profiles/r02_pmc_pss.txt).
//   element n = l + 128 r (lane l, register r)                                     [layout L0, coalesced]
//   A: 32-point transform over r in registers, twiddle W_4096^(l k1)
//   transposition 1 (LDS, both waves): lane = n3 + 4 k1, register = n2            (l = n3 + 4 n2)
//   B: 32-point transform over n2 in registers, twiddle W_128^(n3 k2a)
//   transposition 2: lane = k1 + 32 c, register = d + 8 n3                         (k2a = c + 4 d)
//   C: 4-point transforms over n3 in registers -> X[k1 + 32 k2a + 1024 k2b] in register d + 8 k2b of lane k1 + 32 c: layout L0 again.
// LDS: one 16.9 KB image per block, used for the real and the imaginary parts in turn; addresses chosen so that every wave-wide access
// touches the 64 banks once each (132 = 4 mod 64 floats between the k1 rows of image 1; a rotation inside 64-float rows in image 2).

template <bool INV>
static __device__ __forceinline__ void fft4(cx (&x)[4])
{
  const cx a0 = x[0] + x[2], a1 = x[0] - x[2], a2 = x[1] + x[3], a3 = x[1] - x[3];
  x[0] = a0 + a2;
  x[2] = a0 - a2;
  x[1] = add_rot<INV>(a1, a3);
  x[3] = sub_rot<INV>(a1, a3);
}

template <int N2, int K1, bool INV>
static __device__ __forceinline__ cx inner_twiddle32(cx z)
{
  constexpr int j = (2 * N2 * K1) & 63; // W_32^(n2 k1) = W_64^(2 n2 k1)
  if constexpr (j == 0) {
    return z;
  } else if constexpr (j == 16) {
    return add_rot<INV>(cx{0.f, 0.f}, z);
  } else {
    return cmul2<false>(z, w64<j, INV>());
  }
}
template <int N2, bool INV>
static __device__ __forceinline__ void fft32_column(cx (&a)[32])
{
  cx t[4] = {a[N2], a[8 + N2], a[16 + N2], a[24 + N2]};
  fft4<INV>(t);
  a[N2]      = t[0];
  a[8 + N2]  = inner_twiddle32<N2, 1, INV>(t[1]);
  a[16 + N2] = inner_twiddle32<N2, 2, INV>(t[2]);
  a[24 + N2] = inner_twiddle32<N2, 3, INV>(t[3]);
}
// 32-point transform of 32 registers, natural order in and out (n = 8 n1 + n2, k = k1 + 4 k2)
template <bool INV>
static __device__ __forceinline__ void fft32(cx (&a)[32])
{
  fft32_column<0, INV>(a);
  fft32_column<1, INV>(a);
  fft32_column<2, INV>(a);
  fft32_column<3, INV>(a);
  __builtin_amdgcn_sched_barrier(0);
  fft32_column<4, INV>(a);
  fft32_column<5, INV>(a);
  fft32_column<6, INV>(a);
  fft32_column<7, INV>(a);
  __builtin_amdgcn_sched_barrier(0);
  cx o[32];
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {
    cx t[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; n2++) {
      t[n2] = a[8 * k1 + n2];
    }
    fft8<INV>(t);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
      o[k1 + 4 * k2] = t[k2];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 32; i++) {
    a[i] = o[i];
  }
}

constexpr int kPairImg = 4224; // floats: max(32 * 132, 64 * 64)

// v[r] of lane l = element l + 128 r, in and out.  T1[b] = W^(l b) (b < 8), T2[a] = W^(8 l a) (a < 4), W = e^{-2 pi i / 4096};
// tab[n3 * 32 + k] = W_128^(n3 k) in LDS.
template <bool INV>
static __device__ __forceinline__ void fft4096_pair(cx (&v)[32], const cx (&T1)[8], const cx (&T2)[4], float* img, const cx* tab, int lane)
{
  fft32<INV>(v); // A
#pragma unroll
  for (int k = 1; k < 32; k++) {
    const cx w = (k & 7) == 0 ? T2[k >> 3] : ((k >> 3) == 0 ? T1[k & 7] : cmul2<false>(T2[k >> 3], T1[k & 7]));
    v[k]       = cmul2<INV>(v[k], w);
  }
  __builtin_amdgcn_sched_barrier(0);
  // transposition 1
  const int r1 = (lane >> 2) * 132 + (lane & 3);
#pragma unroll
  for (int k1 = 0; k1 < 32; k1++) {
    img[k1 * 132 + lane] = v[k1].x;
  }
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 32; n2++) {
    v[n2].x = img[r1 + 4 * n2];
  }
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 32; k1++) {
    img[k1 * 132 + lane] = v[k1].y;
  }
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 32; n2++) {
    v[n2].y = img[r1 + 4 * n2];
  }
  __syncthreads();
  fft32<INV>(v); // B
  const cx* tb = tab + (lane & 3) * 32;
#pragma unroll
  for (int k = 1; k < 32; k++) {
    v[k] = cmul2<INV>(v[k], tb[k]);
  }
  __builtin_amdgcn_sched_barrier(0);
  // transposition 2: element (k1, n3, k2a) sits at 64 (w + 2 k2a) + ((k1lo + 16 (n3 + w + 2 c)) mod 64) with k1 = k1lo + 16 w, c = k2a mod 4:
  // the 64 lanes of a wave (16 k1lo x 4 n3 when writing, 16 k1lo x 2 w x 2 c when reading) touch the 64 banks once each
  int wofs[4], rofs[4];
  {
    const int k1 = lane >> 2, n3 = lane & 3, w = k1 >> 4, k1lo = k1 & 15;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      wofs[c] = 64 * w + ((k1lo + 16 * (n3 + w) + 32 * c) & 63);
    }
    const int kr = lane & 31, cr = lane >> 5, wr = kr >> 4, klo = kr & 15;
#pragma unroll
    for (int n3r = 0; n3r < 4; n3r++) {
      rofs[n3r] = 64 * wr + 128 * cr + ((klo + 16 * wr + 32 * cr + 16 * n3r) & 63);
    }
  }
#pragma unroll
  for (int k = 0; k < 32; k++) {
    img[wofs[k & 3] + 128 * k] = v[k].x;
  }
  __syncthreads();
#pragma unroll
  for (int n3 = 0; n3 < 4; n3++) {
#pragma unroll
    for (int d = 0; d < 8; d++) {
      v[d + 8 * n3].x = img[rofs[n3] + 512 * d];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k++) {
    img[wofs[k & 3] + 128 * k] = v[k].y;
  }
  __syncthreads();
#pragma unroll
  for (int n3 = 0; n3 < 4; n3++) {
#pragma unroll
    for (int d = 0; d < 8; d++) {
      v[d + 8 * n3].y = img[rofs[n3] + 512 * d];
    }
  }
  __syncthreads();
  // C
#pragma unroll
  for (int d = 0; d < 8; d++) {
    cx t[4] = {v[d], v[d + 8], v[d + 16], v[d + 24]};
    fft4<INV>(t);
    v[d]      = t[0];
    v[d + 8]  = t[1];
    v[d + 16] = t[2];
    v[d + 24] = t[3];
  }
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 3))) void pss_pair_kernel(const PssParams p)
{
  __shared__ float img[kPairImg];
  __shared__ cx    tab[4 * 32];
  __shared__ float s_best[2];
  __shared__ int   s_besti[2];
  const int cap = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
  const cx* x   = reinterpret_cast<const cx*>(p.in) + (size_t)cap * p.in_stride;
  const cx* tw  = reinterpret_cast<const cx*>(p.twiddle);
  tab[lane]     = tw[(32 * (lane >> 5) * (lane & 31)) & 4095];

  cx T1[8], T2[4];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    T1[j] = tw[lane * j];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    T2[j] = tw[lane * 8 * j];
  }
  const int i0  = blk * p.hop;           // first convolution output of this block
  const int off = i0 - (p.fft_size - 1); // capture index of segment element 0
  // the capture as a raw buffer: elements before its start (negative offsets wrap to huge ones) and behind its end read as zero
  const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<cx*>(x), 0, p.frame_size * 8, 0x00020000);
  const int m_lo = p.fft_size - 1, m_hi = p.fft_size - 1 + p.hop;
  __syncthreads(); // tab
  bool have_spec = false;
  for (int h = 0; h < 3; h++) {
    if (!(p.n_id_2_mask & (1 << h))) {
      continue;
    }
    // (nothing below may look loop-invariant, see pss_wave_kernel)
    int lane_h = lane, m_lo_h = m_lo;
    asm volatile("" : "+v"(lane_h), "+s"(m_lo_h));
#pragma unroll
    for (int j = 0; j < 8; j++) {
      asm volatile("" : "+v"(T1[j].x), "+v"(T1[j].y));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      asm volatile("" : "+v"(T2[j].x), "+v"(T2[j].y));
    }
    cx        v[32], f[32];
    const cx* filt = reinterpret_cast<const cx*>(p.filt) + (size_t)h * 4096 + lane_h;
    cx*       spec = reinterpret_cast<cx*>(p.spec) + ((size_t)cap * p.n_blocks + blk) * 4096 + lane_h;
    if (!have_spec) {
#pragma unroll
      for (int r = 0; r < 32; r++) {
        v[r] = __builtin_bit_cast(cx, __builtin_amdgcn_raw_buffer_load_b64(xb, (off + lane_h + 128 * r) * 8, 0, 0));
      }
      fft4096_pair<false>(v, T1, T2, img, tab, lane_h);
      __builtin_amdgcn_sched_barrier(0);
      if ((p.n_id_2_mask >> (h + 1)) != 0) { // the spectrum waits in global memory for the other hypotheses (see pss_wave_kernel)
#pragma unroll
        for (int r = 0; r < 32; r++) {
          spec[128 * r] = v[r];
        }
      }
      have_spec = true;
    } else {
#pragma unroll
      for (int r = 0; r < 32; r++) {
        v[r] = spec[128 * r];
      }
    }
#pragma unroll
    for (int r = 0; r < 32; r++) {
      f[r] = filt[128 * r];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 32; r++) {
      v[r] = cmul2<false>(v[r], f[r]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      asm volatile("" : "+v"(T1[j].x), "+v"(T1[j].y)); // (no common sub-expressions with the forward transform's twiddle products)
    }
    fft4096_pair<true>(v, T1, T2, img, tab, lane_h);

    const __amdgpu_buffer_rsrc_t cb =
        __builtin_amdgcn_make_buffer_rsrc(p.corr + ((size_t)cap * 3 + h) * p.corr_stride, 0, p.n_out * 4, 0x00020000);
    const bool ema   = p.ema_alpha > 0.0f && p.ema_alpha < 1.0f;
    float      best  = -1.0f;
    int        besti = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 32; r++) {
      if (128 * r + 127 < m_lo_h) { // wave-uniform: the whole register row lies in the discarded head of the block
        continue;
      }
      const int  m = lane_h + 128 * r, i = off + m;
      const bool ok = m >= m_lo_h && m < m_hi && i < p.n_out;
      const int  bo = ok ? i * 4 : -1;
      float      pw = v[r].x * v[r].x + v[r].y * v[r].y; // srsran_vec_abs_square_cf
      if (ema) {
        const float old = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cb, bo, 0, 0));
        pw              = pw * p.ema_alpha + old * (1.0f - p.ema_alpha); // pss.c:497-500
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pw), cb, bo, 0, 0);
      const bool gt = ok && pw > best; // i grows with r: the first maximum of the lane stays
      best          = gt ? pw : best;
      besti         = gt ? i : besti;
    }
    // block arg-max (first maximum wins on ties, as srsran_vec_max_fi)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob  = __shfl_down(best, o);
      const int   obi = __shfl_down(besti, o);
      if (ob > best || (ob == best && obi < besti)) {
        best  = ob;
        besti = obi;
      }
    }
    if ((lane & 63) == 0) {
      s_best[lane >> 6]  = best;
      s_besti[lane >> 6] = besti;
    }
    __syncthreads();
    if (lane == 0) {
      if (s_best[1] > best || (s_best[1] == best && s_besti[1] < besti)) {
        best  = s_best[1];
        besti = s_besti[1];
      }
      const size_t o = ((size_t)cap * 3 + h) * p.n_blocks + blk;
      p.part_val[o]  = best;
      p.part_idx[o]  = besti;
    }
    __syncthreads();
  }
}

#endif // SRSRAN_HIP_WITH_VARIANTS

hipError_t launch_pss_wave_blocks(const PssParams& p, hipStream_t stream)
{
#ifdef SRSRAN_HIP_WITH_VARIANTS
  if (knob(KNOB_PSS_VARIANT) == 1) { // development knob: "pair" = the measured alternative with two waves per block
    hipLaunchKernelGGL(pss_pair_kernel, dim3(p.n_blocks, p.n_cap), dim3(128), 0, stream, p);
    return hipGetLastError();
  }
#endif
  if (knob(KNOB_PSS_VARIANT) == 3) { // "recompute": no parked spectra (A/B against the product; tools/measure/pss_ab.py)
    hipLaunchKernelGGL(pss_wave_kernel<true>, dim3(p.n_blocks, p.n_cap), dim3(64), 0, stream, p);
  } else {
    hipLaunchKernelGGL(pss_wave_kernel<false>, dim3(p.n_blocks, p.n_cap), dim3(64), 0, stream, p);
  }
  return hipGetLastError();
}

} // namespace sync
} // namespace phyhip
