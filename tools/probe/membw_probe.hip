// membw_probe.hip -- access-pattern bandwidth probe for the turbo decoder's workspace traffic (development tool).
// Each wave streams through its own slab like tdec_win_kernel does: `narr` arrays of `nblk` blocks; per block and
// array a lane moves 32 bytes.  Patterns:
//   0: blocked layout, two dwordx4 per lane at [lane*32, +16) (current load_block)
//   1: split layout, two dwordx4 each fully coalesced (1 KB contiguous per instruction)
//   2: row layout, eight dword loads of 256 B contiguous each (load_rows)
//   3: like 2 but the rows are visited in a pseudo-random (QPP-like) order
// mode bit 8 set: writes instead of reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void probe(uint32_t* ws, size_t slab_dwords, int narr, int nblk, int pattern, int wr, int passes, uint32_t* sink)
{
  const int lane = threadIdx.x;
  uint32_t* base = ws + (size_t)blockIdx.x * slab_dwords;
  const size_t AW = (size_t)nblk * 64 * 8;
  uint32_t acc = 0;
  for (int p = 0; p < passes; p++) {
    for (int b = 0; b < nblk; b++) {
      for (int a = 0; a < narr; a++) {
        uint32_t* arr = base + a * AW;
        if (pattern == 0) {
          uint4* q = reinterpret_cast<uint4*>(arr + ((size_t)b * 64 + lane) * 8);
          if (wr) { q[0] = make_uint4(acc, 1, 2, 3); q[1] = make_uint4(4, 5, 6, acc); }
          else { uint4 x = q[0], y = q[1]; acc += x.x + x.w + y.y + y.w; }
        } else if (pattern == 1) {
          uint4* q0 = reinterpret_cast<uint4*>(arr + (((size_t)b * 2 + 0) * 64 + lane) * 4);
          uint4* q1 = reinterpret_cast<uint4*>(arr + (((size_t)b * 2 + 1) * 64 + lane) * 4);
          if (wr) { *q0 = make_uint4(acc, 1, 2, 3); *q1 = make_uint4(4, 5, 6, acc); }
          else { uint4 x = *q0, y = *q1; acc += x.x + x.w + y.y + y.w; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) {
            int row = b * 8 + j;
            if (pattern == 3) row = (int)(((unsigned)row * 263u + 480u * (unsigned)row * (unsigned)row) % (unsigned)(nblk * 8));
            uint32_t* q = arr + (size_t)row * 64 + lane;
            if (wr) *q = acc + j; else acc += *q;
          }
        }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv)
{
  const int n_waves = argc > 1 ? atoi(argv[1]) : 6656, nblk = 48, narr_max = 8;
  const size_t slab = (size_t)narr_max * nblk * 64 * 8;
  uint32_t *ws, *sink;
  hipMalloc(&ws, slab * n_waves * 4);
  hipMalloc(&sink, 64);
  hipMemset(ws, 0, slab * n_waves * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[4] = {"blocked 2x dwordx4 (32B/lane)", "split 2x dwordx4 (coalesced)", "rows 8x dword", "rows 8x dword, permuted"};
  for (int wr = 0; wr < 2; wr++)
    for (int pat = 0; pat < 4; pat++)
      for (int narr : {1, 4}) {
        const int passes = 4;
        hipLaunchKernelGGL(probe, dim3(n_waves), dim3(64), 0, 0, ws, slab, narr, nblk, pat, wr, 1, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(n_waves), dim3(64), 0, 0, ws, slab, narr, nblk, pat, wr, passes, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)n_waves * passes * nblk * narr * 64 * 32;
        printf("%-5s %-32s narr=%d : %7.1f GB/s (%.3f ms)\n", wr ? "write" : "read", names[pat], narr, bytes / ms / 1e6, ms);
      }
  return 0;
}
