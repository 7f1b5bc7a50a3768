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

// Message arithmetic of the three reference decoders (same layered schedule, ldpc_decoder.c:44-104):
//   Pol8  : ldpc_dec_c.c   int8,  messages |.| <= 63,    soft bits +-127 = infinity
//   Pol16 : ldpc_dec_s.c   int16, messages |.| <= 16383, soft bits +-32767 = infinity
//   PolF  : ldpc_dec_f.c   float, no clipping; only IEEE sub / mul / add, evaluated unfused so that the
//                          result is bit-identical to the reference's
struct Pol8 {
  typedef int8_t  T;
  typedef int16_t TS; // soft bits are kept as int16 in LDS: byte-wide LDS writes of 4 lanes into one dword serialise
  typedef int     A;
  static constexpr bool kC2vInLds = false; // see the kernel: the HBM/L2 slab wins over LDS residency through occupancy
  static constexpr bool kIntegerFast = true;
  static constexpr int  kMinWaves    = 1; // 120 VGPRs as it is
  static __device__ __forceinline__ A min_init() { return 127; }
  static __device__ __forceinline__ A v2c(A s, A c) // ldpc_dec_c.c:338-363
  {
    const A x = min(max(s - c, -63), 63);
    const A i = min(max(s, -127), 127); // +-127 propagates as infinity
    return (s >= 127 || s <= -127) ? i : x;
  }
  static __device__ __forceinline__ A mag(A x) { return x < 0 ? -x : x; }
  static __device__ __forceinline__ bool neg(A x) { return x < 0; }
  static __device__ __forceinline__ A scale(A m, int sf, float) { return m * sf / 100; } // :275-278
  static __device__ __forceinline__ A negate(A m) { return -m; }
  static __device__ __forceinline__ A soft(A cn, A v) // :308-315
  {
    // t > 63 -> 127, t < -63 -> -127, written without compares (no VCC round trips): clamp, then push what was clipped out to
    // the "infinity" value 63 + 64
    const A t = cn + v, k = min(max(t, -63), 63);
    return k + (min(max(t - k, -1), 1) << 6);
  }
};

struct Pol16 {
  typedef int16_t T;
  typedef int16_t TS;
  typedef int     A;
  static constexpr bool kC2vInLds = false;
  static constexpr bool kIntegerFast = true;
  // two 6-wave workgroups share a CU (3 waves per SIMD after balancing, 4 on a SIMD while they overlap): above 128 VGPRs
  // the second workgroup does not fit and the decoder runs at half occupancy (75 ms instead of 55 ms)
  static constexpr int  kMinWaves    = 4;
  static __device__ __forceinline__ A min_init() { return 32767; }
  static __device__ __forceinline__ A v2c(A s, A c) // ldpc_dec_s.c:338-363
  {
    if (s >= 32767) {
      return 32767;
    }
    if (s <= -32767) {
      return -32767;
    }
    const A x = s - c;
    return x > 16383 ? 16383 : (x < -16383 ? -16383 : x);
  }
  static __device__ __forceinline__ A mag(A x) { return x < 0 ? -x : x; }
  static __device__ __forceinline__ bool neg(A x) { return x < 0; }
  static __device__ __forceinline__ A scale(A m, int sf, float) { return (A)(int16_t)(m * sf / 100); } // :275-276
  static __device__ __forceinline__ A negate(A m) { return -m; }
  static __device__ __forceinline__ A soft(A cn, A v) // :303-311
  {
    const A t = cn + v, k = min(max(t, -16383), 16383);
    return k + (min(max(t - k, -1), 1) << 14); // 16383 + 16384 = 32767
  }
};

struct PolF {
  typedef float T;
  typedef float TS;
  typedef float A;
  static constexpr bool kC2vInLds = false;
  static constexpr bool kIntegerFast = false;
  static constexpr int  kMinWaves    = 1;
  static __device__ __forceinline__ A min_init() { return INFINITY; }
  static __device__ __forceinline__ A v2c(A s, A c) { return __fsub_rn(s, c); } // ldpc_dec_f.c:172-181
  static __device__ __forceinline__ A mag(A x) { return fabsf(x); }
  static __device__ __forceinline__ bool neg(A x) { return !(x >= 0); } // (v2c >= 0) ? 1 : -1
  static __device__ __forceinline__ A scale(A m, int, float sf) { return __fmul_rn(m, sf); } // :246
  static __device__ __forceinline__ A negate(A m) { return -m; }
  static __device__ __forceinline__ A soft(A cn, A v) { return __fadd_rn(cn, v); } // :278
};

// One layer (base-graph row) for one lifted check node: all operand loads are issued before the first use
// so that a layer costs two memory round trips, not two per edge.
// c2v: the slab of this workgroup (wave-uniform pointer, so that the accesses take the scalar-base form with a 32-bit
// vector offset); coff: this lane's offset inside one edge row of the slab.
// FLOOD: flooded schedule (ldpc_dec_c_flood.c): the messages are updated, the soft bits are left alone (they are rebuilt from the
// channel LLRs once per iteration, see the kernel).
template <int DEG, class POL, bool FLOOD>
__device__ __forceinline__ void layer(typename POL::TS* soft, typename POL::T* c2v, uint32_t coff, int my_edge, int e0, int c, int Z, int sf,
                                      float sf_f, bool active)
{
  typedef typename POL::A  A;
  typedef typename POL::T  T;
  typedef typename POL::TS TS;
  int ed[DEG], idx[DEG];
  A   sb[DEG], co[DEG], v[DEG];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    ed[i] = __builtin_amdgcn_readlane(my_edge, i);
  }
  if (!active) {
    return;
  }
  const uint32_t row0 = (uint32_t)e0 * (uint32_t)Z;
  const uint32_t cb   = (uint32_t)c * (uint32_t)sizeof(TS), Zb = (uint32_t)Z * (uint32_t)sizeof(TS);
  char*          sbase = reinterpret_cast<char*>(soft);
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    // byte offset of soft[col * Z + (c + shift) mod Z]: the wrapped difference is huge when c + shift < Z
    const uint32_t j = cb + (uint32_t)(ed[i] >> 16) * (uint32_t)sizeof(TS);
    idx[i]           = (int)((uint32_t)(ed[i] & 0xffff) * (uint32_t)sizeof(TS) + min(j, j - Zb));
    sb[i]            = *reinterpret_cast<TS*>(sbase + idx[i]);
    co[i]            = c2v[row0 + (uint32_t)i * (uint32_t)Z + coff];
  }
  if constexpr (POL::kIntegerFast) {
    // integer decoders: the same update written for the VALU -- second minimum as a median (v_med3_i32), sign
    // product as the XOR of the messages' sign bits, conditional negation as (m ^ s) - s
    // The reference gives the edge of the FIRST minimum the second minimum.  Comparing magnitudes instead of tracking that
    // position is the same thing: with two equal smallest magnitudes the second minimum equals the minimum.
    int min0 = POL::min_init(), min1 = POL::min_init(), sgn = 0;
    int a[DEG];
#pragma unroll
    for (int i = 0; i < DEG; i++) {
      const int x = POL::v2c(sb[i], co[i]);
      v[i]        = x;
      a[i]        = x < 0 ? -x : x;
      min1        = min(max(a[i], min0), min1); // second smallest of {a, min0, min1}, given min0 <= min1
      min0        = min(a[i], min0);
      sgn ^= x;
    }
    int s0 = POL::scale(min0, sf, sf_f);
    int s1 = POL::scale(min1, sf, sf_f);
    // keep the two scalings (two quarter-rate multiplications each) here: left alone, the compiler sinks them behind the
    // per-edge select and pays them once per edge
    asm volatile("" : "+v"(s0), "+v"(s1));
#pragma unroll
    for (int i = 0; i < DEG; i++) {
      const int mag = (a[i] == min0) ? s1 : s0;
      const int m   = (sgn ^ v[i]) >> 31; // all ones when the product of the OTHER signs is negative
      const int cn  = (mag ^ m) - m;
      c2v[row0 + (uint32_t)i * (uint32_t)Z + coff] = (T)cn;
      if (!FLOOD) {
        *reinterpret_cast<TS*>(sbase + idx[i]) = (TS)POL::soft(cn, v[i]);
      }
    }
    return;
  }
  A    min0 = POL::min_init(), min1 = POL::min_init();
  int  pos = -1;
  bool neg = false;
  // var->check fused with the check-node scan (ldpc_dec_c.c:245-262 and its _s / _f siblings)
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    const A x = POL::v2c(sb[i], co[i]);
    v[i]      = x;
    const A a = POL::mag(x);
    if (a < min0) {
      min1 = min0;
      min0 = a;
      pos  = i;
    } else if (a < min1) {
      min1 = a;
    }
    neg ^= POL::neg(x);
  }
  // check->var and soft-bit update
  const A s0 = POL::scale(min0, sf, sf_f);
  const A s1 = POL::scale(min1, sf, sf_f);
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    const A    mag  = (i == pos) ? s1 : s0;
    const bool sneg = neg ^ POL::neg(v[i]); // sign = product of all signs * own sign (v >= 0 counts as +)
    const A    cn   = sneg ? POL::negate(mag) : mag;
    c2v[row0 + (uint32_t)i * (uint32_t)Z + coff] = (T)cn;
    if (!FLOOD) {
      *reinterpret_cast<TS*>(sbase + idx[i]) = (TS)POL::soft(cn, v[i]);
    }
  }
}

// ES: CRC early stop (decode_crc_c), its own instantiation so that its state costs the plain decoders no registers
template <class POL, bool FLOOD, bool ES>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(ES ? 4 : POL::kMinWaves, 8))) void ldpc_layered_kernel(const Params p)
{
  typedef typename POL::T  T;
  typedef typename POL::TS TS;
  extern __shared__ int8_t lds[];
  const int Z   = p.Z;
  const int t   = threadIdx.x;
  const int cwl = t / Z;
  const int c   = t - cwl * Z;
  const int liftN = p.bgN * Z;
  const int liftK = p.bgK * Z;
  // Check-to-variable messages live in a slab per RESIDENT workgroup slot, not per code word: a workgroup walks
  // over code words (grid-stride) and re-uses its slab, so the slabs of all resident workgroups (~60 MB for BG1
  // Z=384 int8) stay in L2 / Infinity Cache.  Keeping them in LDS instead (121 KB for int8) would allow a single
  // 6-wave workgroup per CU, 2 + 2 + 1 + 1 waves on the four SIMDs; measured 74 ms against 62 ms for the int16
  // decoder that cannot do it.  With only the soft bits in LDS two workgroups share a CU, 3 waves per SIMD.
  const size_t per_cw = (size_t)liftN; // soft bits (elements of TS) in LDS per code word
  TS*          soft   = reinterpret_cast<TS*>(lds) + (size_t)cwl * per_cw;
  T*             c2v  = reinterpret_cast<T*>(p.c2v_ws) + (size_t)blockIdx.x * p.cpb * p.n_edges * Z; // wave-uniform slab base
  const uint32_t coff = (uint32_t)(cwl < p.cpb ? cwl : 0) * (uint32_t)p.n_edges * (uint32_t)Z + (uint32_t)c;
  int* graph = reinterpret_cast<int*>(lds + (((size_t)p.cpb * per_cw * sizeof(TS) + 15) & ~(size_t)15));
  for (int i = t; i < 48 + p.n_edges; i += blockDim.x) {
    graph[i] = i < 48 ? (i <= p.n_layers ? p.row_start[i] : 0) : p.edges[i - 48];
  }
  // flooded schedule: the edges of every variable node in row order (all bgM rows), behind the row description
  int* col_start = graph + 48 + p.n_edges;
  int* col_edges = col_start + 72;
  if (FLOOD) {
    for (int i = t; i < 72 + p.n_col_edges; i += blockDim.x) {
      col_start[i] = i < 72 ? (i <= p.bgN ? p.col_start[i] : 0) : p.col_edges[i - 72];
    }
  }
  const int      sf        = p.sf;
  const float    sf_f      = p.sf_f;
  const int      msg_bytes = (liftK + 7) >> 3;
  // the graph description lives in LDS behind the code-word state (wave-uniform broadcast reads):
  //   row_start[l] : first edge of layer l ;  edges[e] = (col * Z) | shift << 16
  const int* row_start = graph;
  const int* edges     = graph + 48;

  // code words are handed out by a counter when the host provides one (Params::work_counter, see ldpc_packed_kernels.hip): fixed shares wait for the slowest CU
  __shared__ int s_next;
  for (int cw0 = blockIdx.x * p.cpb; cw0 < p.n_cw;) {
  const int  cw     = cw0 + cwl;
  const bool present = (cwl < p.cpb) && (cw < p.n_cw);
  const int  cwi     = (present && p.cw_map) ? (int)p.cw_map[cw] : cw; // row of the LLR / message arrays
  bool       active  = present; // cleared once this code word's CRC matches (early stop): its state is then frozen
  int        it_done = 0;
  __syncthreads(); // the previous code word's message extraction has finished reading soft[]

  // init_ldpc_dec_c (ldpc_dec_c.c:170-188): punctured nodes 0,1 start at 0, all c2v at 0
  if (active) {
    const T* llr = reinterpret_cast<const T*>(p.llrs) + (size_t)cwi * p.llr_stride;
    soft[c]     = 0;
    soft[Z + c] = 0;
    for (int n = 2; n < p.bgN; n++) {
      soft[n * Z + c] = llr[(n - 2) * Z + c];
    }
    for (int e = 0; e < p.n_edges; e++) {
      c2v[(uint32_t)e * (uint32_t)Z + coff] = 0;
    }
  }
  __syncthreads();

  // the description of a layer (first edge, degree, one edge word per lane) is fetched one layer ahead, so that its
  // two dependent LDS round trips overlap the arithmetic of the current layer instead of heading every layer
  const int lane = t & 63;
  int       e0n  = __builtin_amdgcn_readfirstlane(row_start[0]);
  int       degn = __builtin_amdgcn_readfirstlane(row_start[1]) - e0n;
  int       edgn = edges[e0n + (lane < degn ? lane : 0)];
  const int n_iter = FLOOD ? 2 * p.max_iter : p.max_iter; // ldpc_decoder.c:136
  for (int it = 0; it < n_iter; it++) {
    for (int l = 0; l < p.n_layers; l++) {
      // wave-uniform by construction; readfirstlane tells the compiler so (scalar switch, scalar addressing)
      const int e0 = e0n, deg = degn, my_edge = edgn;
      {
        const int ln = l + 1 < p.n_layers ? l + 1 : 0;
        e0n          = __builtin_amdgcn_readfirstlane(row_start[ln]);
        degn         = __builtin_amdgcn_readfirstlane(row_start[ln + 1]) - e0n;
        edgn         = edges[e0n + (lane < degn ? lane : 0)];
      }
      switch (deg) {
#define LDPC_CASE(D)                                                                                                   \
  case D:                                                                                                              \
    layer<D, POL, FLOOD>(soft, c2v, coff, my_edge, e0, c, Z, sf, sf_f, active);                                                     \
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
      if (!FLOOD) {
        __syncthreads(); // the next row reads the soft bits this one wrote
      }
    }
    if (FLOOD) {
      // update_ldpc_soft_bits_c_flood (ldpc_dec_c_flood.c:304-349): soft = channel LLR, then the messages of ALL rows are added in
      // row order with the saturation after every addition (rows that were not decoded contribute zero; the saturation
      // still applies).  Lane = position inside the lifted variable node.
      __syncthreads(); // all messages of this iteration are written
      if (active) {
        const T* llr = reinterpret_cast<const T*>(p.llrs) + (size_t)cwi * p.llr_stride;
        for (int v = 0; v < p.bgN; v++) {
          typename POL::A acc = v < 2 ? 0 : (typename POL::A)llr[(v - 2) * Z + c];
          for (int ce = col_start[v]; ce < col_start[v + 1]; ce++) {
            const int w = col_edges[ce], e = w & 0xffff;
            typename POL::A m = 0;
            if (e < p.n_edges) {
              int ci = c - (w >> 16);
              ci += ci < 0 ? Z : 0;
              m = (typename POL::A)c2v[(uint32_t)e * (uint32_t)Z + (coff - (uint32_t)c) + (uint32_t)ci];
            }
            acc = POL::soft(m, acc);
          }
          soft[v * Z + c] = (TS)acc;
        }
      }
      __syncthreads();
    }
    if (ES && p.crc_order) {
      // CRC early stop (ldpc_decoder.c:87-99, crc.c:187-193): the check "checksum of the first liftK - order bits equals the
      // last order bits" is the remainder of ALL liftK hard decisions being zero.  Every lane takes bgK consecutive bits,
      // shifts its partial remainder into place with x^(bits behind it) mod g, one lane per code word folds them.
      uint32_t* red = reinterpret_cast<uint32_t*>(col_edges + (FLOOD ? p.n_col_edges : 0));
      const uint32_t order = (uint32_t)p.crc_order, mask = order == 32 ? 0xffffffffu : ((1u << order) - 1u), poly = p.crc_poly & mask;
      uint32_t r = 0;
      if (active) {
        for (int k = 0; k < p.bgK; k++) {
          const uint32_t bit = soft[c * p.bgK + k] < 0 ? 1u : 0u;
          r = ((r << 1) & mask) ^ ((((r >> (order - 1)) ^ bit) & 1u) ? poly : 0u);
        }
        const uint32_t m = p.crc_mult[c];
        uint32_t       q = 0;
        for (int i = (int)order - 1; i >= 0; i--) {
          q = ((q << 1) & mask) ^ (((q >> (order - 1)) & 1u) ? poly : 0u);
          q ^= ((m >> i) & 1u) ? r : 0u;
        }
        r = q;
      }
      // fold the partial remainders of a code word.  One code word per workgroup (Z > 128): XOR inside each wave first, one
      // word per wave in LDS (the workgroup's LDS budget at two workgroups per CU has no room for more); several code words
      // per workgroup (small Z): one word per lane, folded by the first lane of each code word.
      uint32_t total;
      if (p.cpb == 1) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          r ^= __shfl_xor(r, off);
        }
        if ((t & 63) == 0) {
          red[t >> 6] = r;
        }
        __syncthreads();
        total = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
          total ^= red[w];
        }
      } else {
        red[t] = r;
        __syncthreads();
        if (active && c == 0) {
          uint32_t x = 0;
          for (int i = 0; i < Z; i++) {
            x ^= red[t + i];
          }
          red[t] = x;
        }
        __syncthreads();
        total = active ? red[t - c] : 1u;
      }
      if (active && total == 0) {
        active  = false;
        it_done = it + 1;
      }
      if (__syncthreads_and(!active)) {
        break; // every code word of this workgroup has stopped (or was never there)
      }
    }
    if (p.iter_msgs) {
      // hard decisions of this iteration, packed MSB first (for the host-side CRC early stop)
      if (present) {
        uint8_t* dst = p.iter_msgs + ((size_t)cw * n_iter + it) * msg_bytes;
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
  if (ES && present && p.n_iter_out) {
    if (c == 0) {
      p.n_iter_out[cw] = it_done; // iterations until the CRC matched; 0: it never did (what decode_crc_c returns)
    }
  }
  if (present) {
    uint8_t* m = p.msg + (size_t)cwi * p.msg_stride;
    for (int i = c; i < liftK; i += Z) {
      m[i] = soft[i] < 0;
    }
    if (p.soft_out) { // parity aid: a-posteriori soft bits
      T* so = reinterpret_cast<T*>(p.soft_out) + (size_t)cw * liftN;
      for (int n = 0; n < p.bgN; n++) {
        so[n * Z + c] = (T)soft[n * Z + c];
      }
    }
  }
  if (p.work_counter) { // (the barrier at the top of the loop keeps s_next from being overwritten before everybody has read it)
    if (threadIdx.x == 0) {
      s_next = (int)((unsigned)gridDim.x * (unsigned)p.cpb + atomicAdd(p.work_counter, (unsigned)p.cpb));
    }
    __syncthreads();
    cw0 = s_next;
  } else {
    cw0 += (int)gridDim.x * p.cpb;
  }
  } // code words of this workgroup
}

size_t lds_bytes(const Params& p);

static size_t elem_size(int dtype)
{
  return dtype == DT_F32 ? 4 : (dtype == DT_I16 ? 2 : 1);
}

// Workgroups launched: exactly as many as are resident at once (CUs x workgroups per CU by waves and LDS), so that
// every slot walks over the same number of code words and no partial round is left at the end (measured on BG1 Z=384:
// 512-768 slots 61 ms, 1024 slots 77 ms, 256 slots 81 ms).
int grid_slots(const Params& p)
{
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int             dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int waves  = (p.cpb * p.Z + 63) / 64;
  int       per_cu = 16 / waves;
  const int by_lds = (int)((160 * 1024) / lds_bytes(p));
  per_cu           = per_cu < by_lds ? per_cu : by_lds;
  per_cu           = per_cu < 1 ? 1 : per_cu;
  int slots        = cus * per_cu;
  if (p.crc_order && !p.work_counter) {
    // early stop with fixed shares: code words take different times, and workgroups queued behind the resident ones even the load out
    // (BG1 Z = 384, 8192 words at 3 iterations on average: 1280 slots 2.08 ms, 1536 1.95 ms, 2048 1.90 ms)
    slots = p.max_slots;
  }
  slots            = slots > p.max_slots ? p.max_slots : slots;
  if (const int v = knob(KNOB_LDPC_SLOTS); v > 0) { // development knob
    slots = v <= p.max_slots ? v : slots;
  }
  const int groups = (p.n_cw + p.cpb - 1) / p.cpb;
  return groups < slots ? groups : slots;
}

size_t lds_bytes(const Params& p)
{
  const size_t per_cw = (size_t)p.bgN * p.Z * (p.dtype == DT_F32 ? 4 : 2); // soft bits: int16 (int8 and int16 decoders) or float
  return (((size_t)p.cpb * per_cw + 15) & ~(size_t)15) + (48 + (size_t)p.n_edges + 72 + (p.flood ? (size_t)p.n_col_edges : 0) + (p.crc_order ? (p.cpb == 1 ? (size_t)16 : (size_t)((p.cpb * p.Z + 63) / 64) * 64 + 8) : 0)) * sizeof(int);
}

template <class POL, bool FLOOD, bool ES>
static hipError_t launch_pol(const Params& p, hipStream_t stream)
{
  const size_t lds = lds_bytes(p);
  static bool  attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ldpc_layered_kernel<POL, FLOOD, ES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); // a few static words (block-wide vote) come on top
    if (e != hipSuccess) {
      return e;
    }
    attr_set = true;
  }
  int threads = p.cpb * p.Z;
  threads     = ((threads + 63) / 64) * 64;
  dim3 grid(grid_slots(p));
  hipLaunchKernelGGL((ldpc_layered_kernel<POL, FLOOD, ES>), grid, dim3(threads), lds, stream, p);
  return hipGetLastError();
}

hipError_t launch(const Params& p, hipStream_t stream)
{
  if (p.packed) {
    return launch_packed(p, stream);
  }
  switch (p.dtype) {
    case DT_I8:
      if (p.crc_order) {
        return p.flood ? launch_pol<Pol8, true, true>(p, stream) : launch_pol<Pol8, false, true>(p, stream);
      }
      return p.flood ? launch_pol<Pol8, true, false>(p, stream) : launch_pol<Pol8, false, false>(p, stream);
    case DT_I16:
      return (p.flood || p.crc_order) ? hipErrorInvalidValue : launch_pol<Pol16, false, false>(p, stream);
    case DT_F32:
      return (p.flood || p.crc_order) ? hipErrorInvalidValue : launch_pol<PolF, false, false>(p, stream);
    default:
      return hipErrorInvalidValue;
  }
}

} // namespace ldpc
} // namespace phyhip
