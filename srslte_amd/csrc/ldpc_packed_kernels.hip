// ldpc_packed_kernels.hip -- the int8 layered LDPC decoder (ldpc_dec_c.c) with TWO lifted positions per lane.
//
// The decoder of ldpc_kernels.hip is bound by VALU issue (about 37 instructions per edge and lifted position),
// and every value it handles fits in 8 bits.  For even lifting sizes Z = 2H this kernel lets lane c (c < H) own the check
// nodes c and c + H of every base-graph row and keeps both messages in one VGPR as int16x2, so that the update runs on
// the packed 16-bit VALU (v_pk_add/sub/min/max/mad/ashr_i16): same arithmetic, bit for bit, half the instructions.
//   * soft bits in LDS: one 16-bit word per (variable node n, p < H) holding positions p (low byte) and p + H (high
//     byte).  The circular shift of an edge sends the lane's pair (c, c + H) to (j, j + H mod Z) with j = (c + s) mod Z:
//     that is the word j mod H, bytes in order when j < H and swapped otherwise -- the swap is folded into the byte
//     selector of the v_perm_b32 that widens the two bytes to int16x2 (and narrows them again on the way back).
//   * check-to-variable messages: one 16-bit word per (edge, lane) in the same per-slot slab as the plain kernel.
//   * scaling x * (int)(sf * 100) / 100 (ldpc_dec_c.c:275-278) as (x * M) >> 9 on the packed 16-bit multiplier, with M
//     checked on the host to give the same quotient for every x in 0..127 (otherwise the plain kernel runs).
// Everything else (layer pipeline, slabs, CRC early stop by the remainder of all K Z hard bits) follows ldpc_kernels.hip.
#include "hip_common.h"
#include "ldpc_device.h"

namespace phyhip {
namespace ldpc {

typedef short          s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ s2v as_s2(uint32_t u)
{
  return __builtin_bit_cast(s2v, u);
}
static __device__ __forceinline__ uint32_t as_u32(s2v v)
{
  return __builtin_bit_cast(uint32_t, v);
}
static __device__ __forceinline__ s2v splat2(short v)
{
  s2v r = {v, v};
  return r;
}
static __device__ __forceinline__ s2v pmin(s2v a, s2v b)
{
  return __builtin_elementwise_min(a, b);
}
static __device__ __forceinline__ s2v pmax(s2v a, s2v b)
{
  return __builtin_elementwise_max(a, b);
}
static __device__ __forceinline__ s2v clip(s2v x, short lim)
{
  return pmin(pmax(x, splat2((short)-lim)), splat2(lim));
}
// KEEP_A: keep the magnitudes in registers between the two passes (the early-stop instantiation recomputes them instead:
// its extra state would push the 19-edge rows into scratch)
template <int DEG, bool KEEP_A>
__device__ __forceinline__ void layer_packed(char* sbase, uint16_t* c2v, uint32_t coff, int my_edge, int e0, int c, uint32_t Z, uint32_t H,
                                             unsigned short m9, bool active)
{
  int ed[DEG];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    ed[i] = __builtin_amdgcn_readlane(my_edge, i);
  }
  if (!active) {
    return;
  }
  const uint32_t coffb = coff * 2u;        // this lane's byte offset inside an edge's run of messages (32-bit: scalar base + VGPR offset addressing)
  const uint32_t rowb  = (uint32_t)e0 * Z; // byte offset of this row's first edge (H words of 2 bytes per edge); wave-uniform
  char*          cbase = reinterpret_cast<char*>(c2v);
  // (byte-wide d16 loads / stores of the two halves would save the widening and narrowing instructions below, but were
  // measured 28 % slower: the LDS and vector-memory pipes then take twice the instructions)
  int  idx[DEG];  // LDS byte offset of the soft word holding the lane's pair
  bool swp[DEG];  // ... byte-swapped in it (a lane mask in scalar registers)
  s2v  s[DEG], co[DEG];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    // lane pair (c, c + H) rotated by the shift: word (c + shift) mod H, bytes swapped when floor((c + shift) / H) is odd
    const uint32_t sh = (uint32_t)ed[i] >> 16;            // scalar, < Z
    const bool     sq = sh >= H;                          // scalar
    const uint32_t a  = (uint32_t)c + (sq ? sh - H : sh); // < 2 H
    const uint32_t dw = min(a, a - H);                    // mod H: the wrapped difference is huge when a < H
    swp[i]            = sq ? (dw == a) : (dw != a);
    idx[i]            = (int)(((uint32_t)ed[i] & 0xffffu) + dw * 2u); // node * Z + 2 dw
    const uint32_t sw = *reinterpret_cast<const uint16_t*>(sbase + idx[i]);
    const uint32_t cw = *reinterpret_cast<const uint16_t*>(cbase + (rowb + (uint32_t)i * Z + coffb));
    // bytes 0 / 1 to the halves of the lane's pair, sign-extended (swapped pairs: byte 1 is the low half)
    s[i]  = as_s2(__builtin_amdgcn_perm(0u, sw, swp[i] ? 0x000c010cu : 0x010c000cu)) >> 8;
    co[i] = as_s2(__builtin_amdgcn_perm(0u, cw, 0x010c000cu)) >> 8;
  }
  s2v      x[DEG], a[KEEP_A ? DEG : 1];
  s2v      min0 = splat2(127), min1 = splat2(127);
  uint32_t sgn  = 0;
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    // ldpc_dec_c.c:338-363: +-127 passes as infinity, everything else is clip(s - c, +-63).  Soft words only hold -63..63
    // and +-127 (the kernel normalises the channel LLRs when it loads them), so h = s - clip(s) is +-64 for the
    // infinite values and 0 otherwise; pushing s - c out by 3 h makes the clip produce +-63, h completes it to +-127.
    const s2v h  = s[i] - clip(s[i], 63);
    const s2v t  = (s[i] - co[i]) + h * splat2(3); // (3, not a power of two: one v_pk_mad_i16 instead of a shift and an add)
    const s2v xv = clip(t, 63) + h;
    x[i]         = xv;
    const s2v av = pmax(xv, splat2(0) - xv);
    if (KEEP_A) {
      a[i] = av;
    }
    min1         = pmin(pmax(av, min0), min1); // second smallest of {a, min0, min1}, given min0 <= min1
    min0         = pmin(av, min0);
    sgn ^= as_u32(xv);
  }
  // :275-278
  const u2v mm = {m9, m9};
  const s2v s0 = __builtin_bit_cast(s2v, (u2v)((__builtin_bit_cast(u2v, min0) * mm) >> 9));
  const s2v s1 = __builtin_bit_cast(s2v, (u2v)((__builtin_bit_cast(u2v, min1) * mm) >> 9));
  // the edge(s) holding the minimum get the second minimum (equal magnitudes: both are the same number):
  // max(s0, s1 - 129 (a - min0)) is s1 where a == min0 and s0 (>= 0 > s1 - 129) elsewhere
  const s2v c1 = s1 + min0 * splat2(129);
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    const s2v av  = KEEP_A ? a[i] : pmax(x[i], splat2(0) - x[i]);
    const s2v mag = pmax(s0, c1 - av * splat2(129)); // 129: a multiply-add, and 129 * 127 + 127 still fits 16 bits
    const s2v m   = as_s2(sgn ^ as_u32(x[i])) >> 15; // all ones where the product of the OTHER signs is negative
    const s2v cn  = (mag ^ m) - m;
    *reinterpret_cast<uint16_t*>(cbase + (rowb + (uint32_t)i * Z + coffb)) = (uint16_t)__builtin_amdgcn_perm(0u, as_u32(cn), 0x0c0c0200u);
    // :308-315: t > 63 -> 127, t < -63 -> -127.  u = clip(t, 64) reaches +-64 exactly when t is out of range, and then
    // u - clip(u, 63) = +-1 (written with a multiplication so that it is not expanded into compares as a signum)
    const s2v u   = clip(cn + x[i], 64);
    const s2v res = u + (u - clip(u, 63)) * splat2(63);
    *reinterpret_cast<uint16_t*>(sbase + idx[i]) = (uint16_t)__builtin_amdgcn_perm(0u, as_u32(res), swp[i] ? 0x0c0c0002u : 0x0c0c0200u);
  }
}

static size_t lds_bytes_packed(const Params& p)
{
  const size_t per_cw = (size_t)p.bgN * (p.Z / 2) * 2;
  const size_t red    = p.crc_order ? (p.cpb == 1 ? (size_t)16 : (size_t)((p.cpb * (p.Z / 2) + 63) / 64) * 64 + 8) : 0;
  return (((size_t)p.cpb * per_cw + 15) & ~(size_t)15) + (48 + (size_t)p.n_edges + red) * sizeof(int);
}

template <bool ES>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 8))) void ldpc_packed_kernel(const Params p)
{
  extern __shared__ int8_t lds[];
  const uint32_t Z = (uint32_t)p.Z, H = Z >> 1;
  const int      t   = threadIdx.x;
  const int      cwl = t / (int)H;
  const int      c   = t - cwl * (int)H;
  const int      liftK = p.bgK * (int)Z;
  const size_t   per_cw = (size_t)p.bgN * H * 2; // bytes of soft words per code word
  char*          sbase  = reinterpret_cast<char*>(lds) + (size_t)cwl * per_cw;
  uint16_t*      soft2  = reinterpret_cast<uint16_t*>(sbase);
  uint16_t*      c2v    = reinterpret_cast<uint16_t*>(p.c2v_ws) + (size_t)blockIdx.x * p.cpb * p.n_edges * H; // wave-uniform slab base
  const uint32_t coff   = (uint32_t)(cwl < p.cpb ? cwl : 0) * (uint32_t)p.n_edges * H + (uint32_t)c;
  int* graph = reinterpret_cast<int*>(lds + (((size_t)p.cpb * per_cw + 15) & ~(size_t)15));
  for (int i = t; i < 48 + p.n_edges; i += blockDim.x) {
    graph[i] = i < 48 ? (i <= p.n_layers ? p.row_start[i] : 0) : p.edges[i - 48];
  }
  const int* row_start = graph;
  const int* edges     = graph + 48;
  const unsigned short m9 = (unsigned short)p.sf_m9;

  for (int cw0 = blockIdx.x * p.cpb; cw0 < p.n_cw; cw0 += gridDim.x * p.cpb) { // uniform trip count per workgroup
    const int  cw      = cw0 + cwl;
    const bool present = (cwl < p.cpb) && (cw < p.n_cw);
    const int  cwi     = (present && p.cw_map) ? (int)p.cw_map[cw] : cw; // row of the LLR / message arrays
    bool       active  = present;
    int        it_done = 0;
    __syncthreads(); // the previous code word's message extraction has finished reading the soft words

    // init_ldpc_dec_c (ldpc_dec_c.c:170-188)
    if (active) {
      const int8_t* llr = reinterpret_cast<const int8_t*>(p.llrs) + (size_t)cwi * p.llr_stride;
      soft2[c]     = 0;
      soft2[H + c] = 0;
      for (int n = 2; n < p.bgN; n++) {
        // values the first variable-to-check pass treats alike are stored alike: |llr| >= 127 is infinity (+-127), anything
        // else enters as clip(llr, +-63) (its check-to-variable messages are still zero then, ldpc_dec_c.c:345-353)
        auto norm = [](int v) { return v >= 127 ? 127 : (v <= -127 ? -127 : (v > 63 ? 63 : (v < -63 ? -63 : v))); };
        const uint32_t lo = (uint8_t)norm(llr[(n - 2) * Z + c]), hi = (uint8_t)norm(llr[(n - 2) * Z + H + c]);
        soft2[n * H + c] = (uint16_t)(lo | (hi << 8));
      }
      for (int e = 0; e < p.n_edges; e++) {
        c2v[(uint32_t)e * H + coff] = 0;
      }
    }
    __syncthreads();

    const int lane = t & 63;
    int       e0n  = __builtin_amdgcn_readfirstlane(row_start[0]);
    int       degn = __builtin_amdgcn_readfirstlane(row_start[1]) - e0n;
    int       edgn = edges[e0n + (lane < degn ? lane : 0)];
    for (int it = 0; it < p.max_iter; it++) {
      for (int l = 0; l < p.n_layers; l++) {
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
    layer_packed<D, !ES>(sbase, c2v, coff, my_edge, e0, c, Z, H, m9, active);                                               \
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
            break;
        }
        __syncthreads(); // the next row reads the soft words this one wrote
      }
      if (ES && p.crc_order) {
        // remainder of all liftK hard decisions (ldpc_decoder.c:87-99, crc.c:187-193): every lane takes 2 bgK consecutive
        // bits of the message, shifts its partial remainder into place with x^(bits behind it) mod g
        uint32_t*      red   = reinterpret_cast<uint32_t*>(graph + 48 + p.n_edges);
        const uint32_t order = (uint32_t)p.crc_order, mask = order == 32 ? 0xffffffffu : ((1u << order) - 1u), poly = p.crc_poly & mask;
        uint32_t       r     = 0;
        if (active) {
          uint32_t i0  = (uint32_t)c * 2u * (uint32_t)p.bgK;
          uint32_t col = i0 / Z, pos = i0 - col * Z;
          for (int k = 0; k < 2 * p.bgK; k++) {
            const uint32_t half = pos >= H ? 1u : 0u;
            const int8_t   v    = reinterpret_cast<const int8_t*>(sbase)[(col * H + pos - half * H) * 2 + half];
            const uint32_t bit  = v < 0 ? 1u : 0u;
            r = ((r << 1) & mask) ^ ((((r >> (order - 1)) ^ bit) & 1u) ? poly : 0u);
            if (++pos == Z) {
              pos = 0;
              col++;
            }
          }
          const uint32_t m = p.crc_mult[2 * c + 1]; // x^((Z - 2 - 2c) bgK) mod g
          uint32_t       q = 0;
          for (int i = (int)order - 1; i >= 0; i--) {
            q = ((q << 1) & mask) ^ (((q >> (order - 1)) & 1u) ? poly : 0u);
            q ^= ((m >> i) & 1u) ? r : 0u;
          }
          r = q;
        }
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
            uint32_t xs = 0;
            for (uint32_t i = 0; i < H; i++) {
              xs ^= red[t + i];
            }
            red[t] = xs;
          }
          __syncthreads();
          total = active ? red[t - c] : 1u;
        }
        if (active && total == 0) {
          active  = false;
          it_done = it + 1;
        }
        if (__syncthreads_and(!active)) {
          break;
        }
      }
    }
    if (ES && present && p.n_iter_out && c == 0) {
      p.n_iter_out[cw] = it_done;
    }
    // extract_ldpc_message_c (:323-336)
    if (present) {
      uint8_t* m   = p.msg + (size_t)cwi * p.msg_stride;
      uint32_t col = 0, pos = (uint32_t)c;
      for (int i = c; i < liftK; i += (int)H) {
        const uint32_t half = pos >= H ? 1u : 0u;
        m[i] = reinterpret_cast<const int8_t*>(sbase)[(col * H + pos - half * H) * 2 + half] < 0;
        pos += H;
        if (pos >= Z) {
          pos -= Z;
          col++;
        }
      }
    }
  }
}

// the packed kernel applies when ...
bool packed_applies(const Params& p)
{
  if (p.dtype != DT_I8 || p.flood || p.iter_msgs || p.soft_out || (p.Z & 1) || p.Z < 8 || p.sf_m9 <= 0) {
    return false;
  }
  if (const char* e = getenv("LDPC_PACKED")) { // development knob: 0 = plain kernel
    return atoi(e) != 0;
  }
  return true;
}

int grid_slots_packed(const Params& p)
{
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int             dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int waves  = (p.cpb * (p.Z / 2) + 63) / 64;
  int       per_cu = 16 / waves;
  const int by_lds = (int)((160 * 1024) / lds_bytes_packed(p));
  per_cu           = per_cu < by_lds ? per_cu : by_lds;
  per_cu           = per_cu < 1 ? 1 : per_cu;
  int slots        = cus * per_cu;
  if (p.crc_order) {
    // early stop: code words take different times, and workgroups queued behind the resident ones even the load out
    // (BG1 Z = 384, 8192 words at 3 iterations on average: 1280 slots 2.08 ms, 1536 1.95 ms, 2048 1.90 ms)
    slots = p.max_slots;
  }
  slots            = slots > p.max_slots ? p.max_slots : slots;
  if (const char* e = getenv("LDPC_SLOTS")) { // development knob
    slots = atoi(e) > 0 && atoi(e) <= p.max_slots ? atoi(e) : slots;
  }
  const int groups = (p.n_cw + p.cpb - 1) / p.cpb;
  return groups < slots ? groups : slots;
}

template <bool ES>
static hipError_t launch_packed_es(const Params& p, hipStream_t stream)
{
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ldpc_packed_kernel<ES>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) {
      return e;
    }
    attr_set = true;
  }
  int threads = p.cpb * (p.Z / 2);
  threads     = ((threads + 63) / 64) * 64;
  hipLaunchKernelGGL((ldpc_packed_kernel<ES>), dim3(grid_slots_packed(p)), dim3(threads), lds_bytes_packed(p), stream, p);
  return hipGetLastError();
}

hipError_t launch_packed(const Params& p, hipStream_t stream)
{
  if (lds_bytes_packed(p) > 156 * 1024 || p.cpb * (p.Z / 2) > 768) {
    return hipErrorInvalidValue;
  }
  return p.crc_order ? launch_packed_es<true>(p, stream) : launch_packed_es<false>(p, stream);
}

} // namespace ldpc
} // namespace phyhip
