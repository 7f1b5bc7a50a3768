// valu_issue_probe.hip -- what a vector instruction costs in issue time on gfx950, packed 16-bit against plain 32-bit forms, at 1 / 2 / 3 / 4
// waves per SIMD (development tool; build: hipcc --offload-arch=gfx950 -O2 -o valu_issue_probe valu_issue_probe.hip).
// DESIGN.md par. 3.3 rests its LDPC floor on "a v_pk_*_i16 occupies its SIMD for 4 cycles, a plain 32-bit op for 2"; this measures it.
// Each kernel runs ITER x 64 instructions of one opcode, on 8 independent register chains (no dependency stalls), one workgroup of
// W x 4 waves per CU on every CU (W waves per SIMD); cycles from s_memtime around the loop (wave 0 of each workgroup), median over workgroups.
// 100 KB of LDS per workgroup keep it at exactly one workgroup per CU, so W waves per SIMD is what runs.
// Also the wall time per instruction per SIMD (HIP events around the launch): independent of what an s_memtime tick is.
// Output: cycles per instruction per WAVE and per SIMD (= per wave / W).  A dependent-chain column gives the result latency.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2000

#define BODY8(OP)                                                                                                                          \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5)  \
               OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3)  \
               OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])                             \
               : "v"(c))
#define BODY1(OP)                                                                                                                          \
  asm volatile(OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)  \
               OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)  \
               OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)                \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])                             \
               : "v"(c))

#define OP_PK_ADD(i) "v_pk_add_i16 %" #i ", %" #i ", %8 clamp\n"
#define OP_PK_MAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_MIN(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define OP_PK_MAD(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %8\n"
#define OP_PK_MUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define OP_ADD32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_MAX32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define OP_MIN3(i) "v_min3_i32 %" #i ", %" #i ", %8, %8\n"
#define OP_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %8\n"
#define OP_ADD16(i) "v_add_i16 %" #i ", %" #i ", %8 clamp\n"
#define OP_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_DPPROW(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:4 row_mask:0xf bank_mask:0xa\n"
#define OP_SWZ(i) "ds_swizzle_b32 %" #i ", %" #i " offset:swizzle(BITMASK_PERM, \"00p00\")\n"
#define OP_MIN32(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define OP_SUB32(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define OP_AND32(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP_XOR32(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %8\n"
#define OP_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"

#define KERNEL(name, OP, BODY)                                                   \
  __global__ void name(unsigned long long* out, uint32_t seed)                   \
  {                                                                              \
    extern __shared__ uint32_t lds_pad[]; /* 100 KB requested at launch: exactly ONE workgroup per CU */ \
    uint32_t r[8];                                                               \
    for (int i = 0; i < 8; i++) r[i] = seed * (threadIdx.x + i + 1);             \
    if (seed == 0xffffffffu) lds_pad[threadIdx.x] = seed;                        \
    uint32_t c = seed | 1u;                                                      \
    __syncthreads();                                                             \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
    for (int it = 0; it < ITER; it++) {                                          \
      BODY(OP);                                                                  \
    }                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
    uint32_t a = 0;                                                              \
    for (int i = 0; i < 8; i++) a ^= r[i];                                       \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (a == 0x1234567u ? 1 : 0); \
  }

KERNEL(k_pk_add, OP_PK_ADD, BODY8)
KERNEL(k_pk_max, OP_PK_MAX, BODY8)
KERNEL(k_pk_min, OP_PK_MIN, BODY8)
KERNEL(k_pk_mad, OP_PK_MAD, BODY8)
KERNEL(k_pk_mul, OP_PK_MUL, BODY8)
KERNEL(k_add32, OP_ADD32, BODY8)
KERNEL(k_max32, OP_MAX32, BODY8)
KERNEL(k_min3, OP_MIN3, BODY8)
KERNEL(k_med3, OP_MED3, BODY8)
KERNEL(k_add16, OP_ADD16, BODY8)
KERNEL(k_perm, OP_PERM, BODY8)
KERNEL(k_andor, OP_ANDOR, BODY8)
KERNEL(k_cndmask, OP_CNDMASK, BODY8)
KERNEL(k_dpp, OP_DPP, BODY8)
KERNEL(k_dpprow, OP_DPPROW, BODY8)
KERNEL(k_swz, OP_SWZ, BODY8)
KERNEL(k_fma, OP_FMA, BODY8)
KERNEL(k_min32, OP_MIN32, BODY8)
KERNEL(k_sub32, OP_SUB32, BODY8)
KERNEL(k_and32, OP_AND32, BODY8)
KERNEL(k_xor32, OP_XOR32, BODY8)
KERNEL(k_lshl, OP_LSHL, BODY8)
KERNEL(k_bfi, OP_BFI, BODY8)
KERNEL(k_cnd64, OP_CND64, BODY8)
KERNEL(k_add3, OP_ADD3, BODY8)
KERNEL(k_mov, OP_MOV, BODY8)
KERNEL(d_pk_add, OP_PK_ADD, BODY1)
KERNEL(d_add32, OP_ADD32, BODY1)
KERNEL(d_dpp, OP_DPP, BODY1)
KERNEL(d_swz, OP_SWZ, BODY1)
KERNEL(d_cndmask, OP_CNDMASK, BODY1)

typedef void (*kern_t)(unsigned long long*, uint32_t);

int main()
{
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned long long* d = nullptr;
  hipMalloc(&d, sizeof(unsigned long long) * cus);
  struct { const char* name; kern_t k; } ks[] = {
      {"v_pk_add_i16 clamp", k_pk_add}, {"v_pk_max_i16", k_pk_max}, {"v_pk_min_i16", k_pk_min}, {"v_pk_mad_u16", k_pk_mad}, {"v_pk_mul_lo_u16", k_pk_mul},
      {"v_add_u32", k_add32}, {"v_max_i32", k_max32}, {"v_min3_i32", k_min3}, {"v_med3_i32", k_med3}, {"v_add_i16 clamp", k_add16}, {"v_perm_b32", k_perm},
      {"v_and_or_b32", k_andor}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32_dpp quad_perm", k_dpp}, {"v_mov_b32_dpp row_shr:4 bank 0xa", k_dpprow},
      {"ds_swizzle_b32", k_swz}, {"v_fma_f32", k_fma}, {"v_min_i32", k_min32}, {"v_sub_u32", k_sub32}, {"v_and_b32", k_and32}, {"v_xor_b32", k_xor32}, {"v_lshlrev_b32", k_lshl}, {"v_bfi_b32", k_bfi}, {"v_cndmask_b32_e64 (sgpr pair)", k_cnd64}, {"v_add3_u32", k_add3}, {"v_mov_b32", k_mov},
      {"DEPENDENT v_pk_add_i16", d_pk_add}, {"DEPENDENT v_add_u32", d_add32}, {"DEPENDENT v_mov_b32_dpp", d_dpp}, {"DEPENDENT ds_swizzle_b32", d_swz},
      {"DEPENDENT v_cndmask_b32", d_cndmask}};
  printf("%-36s %28s   %28s\n", "instruction (independent x8 chains)", "cycles per instr per WAVE @1/2/3/4 waves per SIMD", "per SIMD (wave / W)");
  for (auto& e : ks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    double pw[4], ns[4];
    for (int w = 1; w <= 4; w++) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.k, dim3(cus), dim3(64 * 4 * w), 100 * 1024, 0, d, 12345u + rep);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      ns[w - 1] = ms * 1e6 / (ITER * 64.0) / w; // wall nanoseconds per instruction per SIMD (W waves of ITER x 64 instructions each)
      std::vector<unsigned long long> h(cus);
      hipMemcpy(h.data(), d, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      pw[w - 1] = (double)h[cus / 2] / (ITER * 64.0);
    }
    printf("%-36s %6.2f %6.2f %6.2f %6.2f        %6.2f %6.2f %6.2f %6.2f      wall ns/instr/SIMD %5.2f %5.2f %5.2f %5.2f\n", e.name, pw[0], pw[1], pw[2], pw[3], pw[0],
           pw[1] / 2, pw[2] / 3, pw[3] / 4, ns[0], ns[1], ns[2], ns[3]);
  }
  hipFree(d);
  return 0;
}
