// ldpc_kernels.hip -- NR LDPC BG1/BG2 layered normalised min-sum decoder (int8) for gfx950.
//
// Bit-exact with the reference int8 layered decoders (scalar lib/src/phy/fec/ldpc/ldpc_dec_c.c and its
// AVX2/AVX512 siblings, which produce identical messages) under the schedule of
// lib/src/phy/fec/ldpc/ldpc_decoder.c:44-104.
//
// MI355X mapping:
//   * one code word per workgroup slot; lane c owns lifted check node c of the current layer (Z lanes);
//     the a-posteriori soft bits (bgN*Z int8) and the check-to-variable messages live in LDS for the
//     whole decode -- HBM sees the LLRs once on the way in and the message once on the way out.
//   * c2v messages are stored per base-graph EDGE and indexed by check node (n_edges*Z int8, 121 KB for
//     BG1 Z=384) instead of the reference's per-layer slab indexed by variable node
//     ((hrrN+Z)*bgM = 477 KB): that is what makes the state fit the 160 KB LDS of a CU.
//   * the circular shift of a base-graph entry is address arithmetic on the LDS read ((c+shift) mod Z),
//     so a layer needs no data movement; min1/min2/argmin/sign are per-lane scalars because the edges of
//     a check are spread over base nodes, not over lanes.
//   * small lifting sizes pack several code words into one workgroup.
#include "hip_common.h"
#include "ldpc_device.h"

namespace phyhip {
namespace ldpc {

#define LDPC_MAX_DEG 19 // BG1 rows 0-3; BG2 max is 10

// One layer (base-graph row) for one lifted check node: all operand loads are issued before the first use
// so that a layer costs two LDS round trips, not two per edge.
template <int DEG>
__device__ __forceinline__ void layer(int8_t* soft, int8_t* c2v, int my_edge, int e0, int c, int Z, int sf, bool active)
{
  int ed[DEG], idx[DEG], sb[DEG], co[DEG], v[DEG];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    ed[i] = __builtin_amdgcn_readlane(my_edge, i);
  }
  if (!active) {
    return;
  }
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    int j  = c + (ed[i] >> 16);
    j      = j >= Z ? j - Z : j;
    idx[i] = (ed[i] & 0xffff) + j;
    sb[i]  = soft[idx[i]];
    co[i]  = c2v[(e0 + i) * Z + c];
  }
  int min0 = 127, min1 = 127, pos = -1, neg = 0;
  // var->check (ldpc_dec_c.c:338-363) fused with the check-node scan (:245-262)
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    int x;
    if (sb[i] >= 127) {
      x = 127;
    } else if (sb[i] <= -127) {
      x = -127;
    } else {
      x = sb[i] - co[i];
      x = x > 63 ? 63 : (x < -63 ? -63 : x);
    }
    v[i]        = x;
    const int a = x < 0 ? -x : x;
    if (a < min0) {
      min1 = min0;
      min0 = a;
      pos  = i;
    } else if (a < min1) {
      min1 = a;
    }
    neg ^= (x < 0);
  }
  // check->var (:265-281) and soft-bit update (:286-321)
  const int s0 = min0 * sf / 100;
  const int s1 = min1 * sf / 100;
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    const int mag  = (i == pos) ? s1 : s0;
    const int sneg = neg ^ (v[i] < 0); // sign = product of all signs * own sign (v >= 0 counts as +)
    const int cn   = sneg ? -mag : mag;
    c2v[(e0 + i) * Z + c] = (int8_t)cn;
    int tt = cn + v[i];
    tt     = tt > 63 ? 127 : (tt < -63 ? -127 : tt);
    soft[idx[i]] = (int8_t)tt;
  }
}

__global__ __launch_bounds__(384) void ldpc_layered_kernel(const Params p)
{
  extern __shared__ int8_t lds[];
  const int Z   = p.Z;
  const int t   = threadIdx.x;
  const int cwl = t / Z;
  const int c   = t - cwl * Z;
  const int cw  = blockIdx.x * p.cpb + cwl;
  const bool active = (cwl < p.cpb) && (cw < p.n_cw);

  const int liftN = p.bgN * Z;
  const int liftK = p.bgK * Z;
  int8_t*   soft  = lds + (size_t)cwl * (liftN + p.n_edges * Z);
  int8_t*   c2v   = soft + liftN;
  int*      graph = reinterpret_cast<int*>(lds + (((size_t)p.cpb * (liftN + p.n_edges * Z) + 15) & ~(size_t)15));
  for (int i = t; i < 48 + p.n_edges; i += blockDim.x) {
    graph[i] = i < 48 ? (i <= p.n_layers ? p.row_start[i] : 0) : p.edges[i - 48];
  }

  // init_ldpc_dec_c (ldpc_dec_c.c:170-188): punctured nodes 0,1 start at 0, all c2v at 0
  if (active) {
    const int8_t* llr = p.llrs + (size_t)cw * p.llr_stride;
    soft[c]     = 0;
    soft[Z + c] = 0;
    for (int n = 2; n < p.bgN; n++) {
      soft[n * Z + c] = llr[(n - 2) * Z + c];
    }
    for (int e = 0; e < p.n_edges; e++) {
      c2v[e * Z + c] = 0;
    }
  }
  __syncthreads();

  const int      sf        = p.sf;
  const int      msg_bytes = (liftK + 7) >> 3;
  // the graph description lives in LDS behind the code-word state (wave-uniform broadcast reads):
  //   row_start[l] : first edge of layer l ;  edges[e] = (col * Z) | shift << 16
  const int* row_start = graph;
  const int* edges     = graph + 48;

  for (int it = 0; it < p.max_iter; it++) {
    for (int l = 0; l < p.n_layers; l++) {
      // wave-uniform by construction; readfirstlane tells the compiler so (scalar switch, scalar addressing)
      const int e0  = __builtin_amdgcn_readfirstlane(row_start[l]);
      const int deg = __builtin_amdgcn_readfirstlane(row_start[l + 1]) - e0;
      // one LDS read per wave fetches the whole row description; v_readlane broadcasts it into SGPRs
      const int lane   = t & 63;
      const int my_edge = edges[e0 + (lane < deg ? lane : 0)];
      switch (deg) {
#define LDPC_CASE(D)                                                                                                   \
  case D:                                                                                                              \
    layer<D>(soft, c2v, my_edge, e0, c, Z, sf, active);                                                                \
    break;
        LDPC_CASE(1)
        LDPC_CASE(2)
        LDPC_CASE(3)
        LDPC_CASE(4)
        LDPC_CASE(5)
        LDPC_CASE(6)
        LDPC_CASE(7)
        LDPC_CASE(8)
        LDPC_CASE(9)
        LDPC_CASE(10)
        LDPC_CASE(19)
#undef LDPC_CASE
        default:
          break; // no such row degree in BG1/BG2 (host checks)
      }
      __syncthreads();
    }
    if (p.iter_msgs) {
      // hard decisions of this iteration, packed MSB first (for the host-side CRC early stop)
      if (active) {
        uint8_t* dst = p.iter_msgs + ((size_t)cw * p.max_iter + it) * msg_bytes;
        for (int b = c; b < msg_bytes; b += Z) {
          uint32_t byte = 0;
          for (int k = 0; k < 8; k++) {
            int i = b * 8 + k;
            if (i < liftK && soft[i] < 0) {
              byte |= 0x80u >> k;
            }
          }
          dst[b] = (uint8_t)byte;
        }
      }
      // soft is only read here; the next layer's writes are ordered by the barrier at its end, but the
      // reads above must finish before layer 0 of the next iteration writes -> barrier
      __syncthreads();
    }
  }
  // extract_ldpc_message_c (:323-336)
  if (active) {
    uint8_t* m = p.msg + (size_t)cw * p.msg_stride;
    for (int i = c; i < liftK; i += Z) {
      m[i] = soft[i] < 0;
    }
  }
}

size_t lds_bytes(const Params& p)
{
  return (((size_t)p.cpb * (size_t)(p.bgN + p.n_edges) * p.Z + 15) & ~(size_t)15) + (48 + (size_t)p.n_edges) * sizeof(int);
}

hipError_t launch(const Params& p, hipStream_t stream)
{
  const size_t lds = lds_bytes(p);
  static bool  attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ldpc_layered_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      return e;
    }
    attr_set = true;
  }
  int threads = p.cpb * p.Z;
  threads     = ((threads + 63) / 64) * 64;
  dim3 grid((p.n_cw + p.cpb - 1) / p.cpb);
  hipLaunchKernelGGL(ldpc_layered_kernel, grid, dim3(threads), lds, stream, p);
  return hipGetLastError();
}

} // namespace ldpc
} // namespace phyhip
