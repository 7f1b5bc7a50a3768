// acs_layout_probe.hip -- VALU cost of the backward (beta) recursion of the LTE turbo trellis in the two lane mappings discussed in
// DESIGN.md par. 3.2 (development tool; build: hipcc --offload-arch=gfx950 -O3 -o acs_layout_probe acs_layout_probe.hip):
//   A  product mapping (turbo_kernels.hip): a lane owns TWO sub-blocks packed as int16x2 and all 8 states in 8 VGPRs; one wave =
//      8 code blocks of 16 sub-blocks; an ACS step is 12 saturating packed adds + 8 packed max, no cross-lane traffic.
//   B  one code block per wave: lane = (sub-block 0..15, state pair q 0..3), int16x2 = states (2q, 2q+1); the two predecessor
//      metrics come from the quad through DPP (quad_perm) + a byte permute that broadcasts the wanted half; the four gamma
//      combinations are selected per lane with v_perm_b32.
// Operands come from registers (a cheap LCG), so the figure is the pure issue cost; both variants run the same number of
// code-block steps.  Output: ns per code-block step and the ratio.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef short s2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ s2 from_u(uint32_t u) { return __builtin_bit_cast(s2, u); }
static __device__ __forceinline__ uint32_t to_u(s2 v) { return __builtin_bit_cast(uint32_t, v); }
static __device__ __forceinline__ s2 adds(s2 a, s2 b) { return __builtin_elementwise_add_sat(a, b); }
static __device__ __forceinline__ s2 vmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }

__global__ __launch_bounds__(64) void probe_a(int steps, uint32_t* sink)
{
  s2 o[8];
  for (int i = 0; i < 8; i++) o[i] = from_u(0x00010001u * (threadIdx.x + i));
  uint32_t r = threadIdx.x * 2654435761u + blockIdx.x;
  for (int k = 0; k < steps; k++) {
    r = r * 1664525u + 1013904223u;
    const s2 x = from_u(r & 0x00ff00ffu), y = from_u((r >> 8) & 0x00ff00ffu);
    const s2 xy = adds(x, y);
    s2 n0 = vmax(adds(o[4], xy), o[0]), n1 = vmax(o[4], adds(o[0], xy));
    s2 n2 = vmax(adds(o[5], y), adds(o[1], x)), n3 = vmax(adds(o[5], x), adds(o[1], y));
    s2 n4 = vmax(adds(o[6], x), adds(o[2], y)), n5 = vmax(adds(o[6], y), adds(o[2], x));
    s2 n6 = vmax(o[7], adds(o[3], xy)), n7 = vmax(adds(o[7], xy), o[3]);
    o[0] = n0; o[1] = n1; o[2] = n2; o[3] = n3; o[4] = n4; o[5] = n5; o[6] = n6; o[7] = n7;
    if ((k & 7) == 7) { // normalisation every 8 steps (turbodecoder_win.h: subtract the metric of state 0)
      const s2 z = o[0];
      for (int i = 0; i < 8; i++) o[i] = o[i] - z;
    }
  }
  uint32_t acc = 0;
  for (int i = 0; i < 8; i++) acc ^= to_u(o[i]);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int CTRL>
static __device__ __forceinline__ uint32_t quad(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

__global__ __launch_bounds__(64) void probe_b(int steps, uint32_t* sink)
{
  const int q = threadIdx.x & 3;
  // byte selectors of v_perm_b32(hi = z, lo = w): bytes 0..3 = w, 4..7 = z, 0x0c = zero; w = (x, y), z = (xy, xy)
  const uint32_t selA = q == 0 ? 0x0c0c0504u : q == 1 ? 0x01000302u : q == 2 ? 0x03020100u : 0x05040c0cu;
  const uint32_t selB = q == 0 ? 0x05040c0cu : q == 1 ? 0x03020100u : q == 2 ? 0x01000302u : 0x0c0c0504u;
  const uint32_t half = (q & 1) ? 0x03020302u : 0x01000100u; // broadcast the half of the predecessor lane's pair this lane needs
  uint32_t o = 0x00010001u * threadIdx.x;
  uint32_t r = (threadIdx.x >> 2) * 2654435761u + blockIdx.x;
  for (int k = 0; k < steps; k++) {
    r = r * 1664525u + 1013904223u;
    const uint32_t w  = r & 0x00ff00ffu;                                   // (x, y) of this sub-block and step
    const uint32_t ws = __builtin_amdgcn_alignbyte(w, w, 2);               // (y, x)
    const uint32_t z  = to_u(adds(from_u(w), from_u(ws)));                 // (x + y, x + y)
    const s2 gA = from_u(__builtin_amdgcn_perm(z, w, selA)), gB = from_u(__builtin_amdgcn_perm(z, w, selB));
    const uint32_t pa = quad<0xfa>(o), pb = quad<0x50>(o);                  // quad_perm [2,2,3,3] and [0,0,1,1]
    const s2 A = from_u(__builtin_amdgcn_perm(0u, pa, half)), B = from_u(__builtin_amdgcn_perm(0u, pb, half));
    o = to_u(vmax(adds(A, gA), adds(B, gB)));
    if ((k & 7) == 7) {
      const uint32_t z0 = __builtin_amdgcn_perm(0u, quad<0x00>(o), 0x01000100u); // state 0 of this sub-block
      o = to_u(from_u(o) - from_u(z0));
    }
  }
  if (o == 0x12345678u) sink[0] = o;
}

int main(int argc, char** argv)
{
  const int steps = 384 * 64, cbs = 65536; // code blocks of 16 sub-blocks
  uint32_t* sink;
  hipMalloc(&sink, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms[2];
  for (int v = 0; v < 2; v++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(probe_a, dim3(cbs / 8), dim3(64), 0, 0, steps, sink);
      else hipLaunchKernelGGL(probe_b, dim3(cbs), dim3(64), 0, 0, steps, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[v], e0, e1);
    }
  }
  const double nA = ms[0] * 1e6 / ((double)cbs * steps), nB = ms[1] * 1e6 / ((double)cbs * steps);
  printf("beta recursion, %d code blocks x %d steps, operands from registers\n", cbs, steps);
  printf("A  product mapping (2 sub-blocks per lane, 8 states in 8 VGPRs, 8 code blocks per wave): %8.3f ms = %.4f ns per code-block step\n", ms[0], nA);
  printf("B  one code block per wave (lane = sub-block x state pair, DPP quad exchange)            : %8.3f ms = %.4f ns per code-block step\n", ms[1], nB);
  printf("B / A = %.2f\n", nB / nA);
  return 0;
}
