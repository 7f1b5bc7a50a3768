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
//   * check-to-variable messages: one 16-bit word per (edge, lane) in the same per-slot slab as the plain kernel, accessed
//     with buffer instructions (descriptor + lane offset + scalar edge offset).
//   * stored bytes are BIASED by 128 (zero extension is then the widening; the biases cancel in s - c) and the reference's
//     +-127 "infinity" is kept as +-64 in the soft words (what a saturating store produces by itself): layer_packed().
//   * the base graph is read through the scalar cache (s_load): no vector instruction and no LDS traffic for it.
//   * scaling x * (int)(sf * 100) / 100 (ldpc_dec_c.c:275-278) as (x * M) >> 9 on the packed 16-bit multiplier, with M
//     checked on the host to give the same quotient for every x in 0..127 (otherwise the plain kernel runs).
// 32 vector instructions per edge and pair of positions (round 1: 43; the steps are in DESIGN 3.3).
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
constexpr short kBias = 128; // stored bytes are value + 128 (layer_packed)
// KEEP_A: keep the magnitudes in registers between the two passes (the early-stop instantiation recomputes them instead:
// its extra state would push the 19-edge rows into scratch)
// CL: the check-to-variable words of this code word are in LDS at byte address c2v_lds (+ edge * Z + 2 c) instead of the global slab
template <int DEG, bool KEEP_A, bool CL>
__device__ __forceinline__ void layer_packed(const __attribute__((address_space(4))) int* ec, char* sbase, __amdgpu_buffer_rsrc_t c2v, uint32_t coff, int e0, int c, uint32_t Z, uint32_t H,
                                             unsigned short m9, bool active, uint32_t c2v_lds)
{
  int ed[DEG];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    ed[i] = ec[e0 + i]; // scalar loads (the graph sits in the constant address space): no vector instruction, no LDS traffic
  }
  if (!active) {
    return;
  }
  const uint32_t coffb = coff * 2u;        // this lane's byte offset inside an edge's run of messages
  const uint32_t rowb  = (uint32_t)e0 * Z; // byte offset of this row's first edge (H words of 2 bytes per edge); wave-uniform
  // Check-to-variable words go through buffer instructions: slab descriptor + this lane's offset (VGPR) + the edge's offset (SGPR),
  // so that no vector instruction is spent on their addresses.
  // Both the soft words and the check-to-variable words are kept BIASED (value + 128 in each byte, see kBias): a byte then widens to
  // its int16 half with the one v_perm_b32 that also sorts the halves (zero extension, no arithmetic shift), the biases cancel in
  // s - c, and the bias of what is stored comes for free with the multiply-add that applies the sign.
  // (byte-wide d16 loads / stores of the two halves would save the widening and narrowing instructions below, but were
  // measured 28 % slower: the LDS and vector-memory pipes then take twice the instructions)
  const uint32_t cb = (uint32_t)(sbase - static_cast<char*>(nullptr)) + (uint32_t)c * 2u; // LDS byte address of (node 0, word c) of this lane's code word
  uint32_t       c2 = (uint32_t)c * 2u, negZ = 0u - Z;
  asm volatile("" : "+v"(c2), "+v"(negZ)); // (values of their own in VGPRs: the select below wants a register operand)
  int  idx[DEG];  // LDS byte address of the soft word holding the lane's pair
  uint32_t wsel[DEG]; // byte selector that narrows the lane's pair back into that word (swapped or not; kept per edge in a VGPR -- the
                      // lane masks of 19 edges do not fit the scalar registers)
  s2v  s[DEG], co[DEG];
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    // lane pair (c, c + H) rotated by the shift: word (c + shift) mod H, bytes swapped when floor((c + shift) / H) is odd.
    // Three vector instructions: does 2 c reach the (scalar) wrap point, select 0 / -Z, three-operand add.
    const uint32_t sh   = (uint32_t)ed[i] >> 16;                  // scalar, < Z
    const bool     sq   = sh >= H;                                // scalar
    const uint32_t sh2  = 2u * (sq ? sh - H : sh);                // scalar: byte offset of the rotation inside the row, < Z
    const bool     wrap = c2 >= Z - sh2;
    const bool swp      = sq != wrap;
    wsel[i]             = swp ? 0x0c0c0002u : 0x0c0c0200u;
    idx[i]              = (int)(cb + (wrap ? negZ : 0u) + (((uint32_t)ed[i] & 0xffffu) + sh2));
    const uint32_t sw   = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>((uintptr_t)idx[i]);
    const uint32_t cw   = CL ? (uint32_t)*reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>((uintptr_t)(c2v_lds + coffb + rowb + (uint32_t)i * Z))
                             : (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(c2v, coffb, rowb + (uint32_t)i * Z, 0);
    // bytes 0 / 1 to the halves of the lane's pair, zero-extended (swapped pairs: byte 1 is the low half)
    s[i]  = as_s2(__builtin_amdgcn_perm(0u, sw, swp ? 0x0c000c01u : 0x0c010c00u));
    co[i] = as_s2(__builtin_amdgcn_perm(0u, cw, 0x0c010c00u));
  }
  short k64s = 64;
  asm volatile("" : "+s"(k64s));
  const s2v k64 = splat2(k64s);
  s2v       x[DEG], a[KEEP_A ? DEG : 1];
  s2v       min0 = splat2(127), min1 = splat2(127);
  uint32_t  sgn  = 0;
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    // ldpc_dec_c.c:338-363: +-127 passes as infinity, everything else is clip(s - c, +-63).  Soft words hold -63..63, and +-64
    // for the reference's +-127 (see the store below), so h = s - clip(s) is +-1 for the infinite values and 0 otherwise;
    // pushing s - c out by 129 h makes the clip produce +-63, 64 h completes it to +-127.
    const s2v h  = s[i] - pmin(pmax(s[i], splat2(kBias - 63)), splat2(kBias + 63));
    const s2v t  = (s[i] - co[i]) + h * splat2(129);
    const s2v xv = clip(t, 63) + h * k64; // (k64 is opaque: one v_pk_mad_u16 instead of a shift and an add)
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
  short kbs = kBias;
  asm volatile("" : "+s"(kbs));
  const s2v kb = splat2(kbs);
#pragma unroll
  for (int i = 0; i < DEG; i++) {
    const s2v av  = KEEP_A ? a[i] : pmax(x[i], splat2(0) - x[i]);
    const s2v mag = pmax(s0, c1 - av * splat2(129)); // 129: a multiply-add, and 129 * 127 + 127 still fits 16 bits
    const s2v sg  = (as_s2(sgn ^ as_u32(x[i])) >> 15) | splat2(1); // -1 where the product of the OTHER signs is negative, else 1
    const s2v cnb = mag * sg + kb;                                 // the new message, biased
    if (CL) {
      *reinterpret_cast<__attribute__((address_space(3))) uint16_t*>((uintptr_t)(c2v_lds + coffb + rowb + (uint32_t)i * Z)) =
          (uint16_t)__builtin_amdgcn_perm(0u, as_u32(cnb), 0x0c0c0200u);
    } else {
      __builtin_amdgcn_raw_buffer_store_b16((uint16_t)__builtin_amdgcn_perm(0u, as_u32(cnb), 0x0c0c0200u), c2v, coffb, rowb + (uint32_t)i * Z, 0);
    }
    // :308-315: t > 63 -> 127, t < -63 -> -127, kept in the soft words as +-64 (only this kernel reads them, and only their sign leaves it)
    const s2v res = pmin(pmax(cnb + x[i], splat2(kBias - 64)), splat2(kBias + 64));
    *reinterpret_cast<__attribute__((address_space(3))) uint16_t*>((uintptr_t)idx[i]) =
        (uint16_t)__builtin_amdgcn_perm(0u, as_u32(res), wsel[i]);
  }
}

static size_t lds_bytes_packed(const Params& p)
{
  const size_t per_cw = (size_t)p.bgN * (p.Z / 2) * 2;
  const size_t red    = p.crc_order ? (p.cpb == 1 ? (size_t)16 : (size_t)((p.cpb * (p.Z / 2) + 63) / 64) * 64 + 8) : 0;
  const size_t c2v    = p.c2v_lds ? (((size_t)p.n_edges * p.Z + 15) & ~(size_t)15) : 0; // behind the soft words and the reduction words
  return (((size_t)p.cpb * per_cw + 15) & ~(size_t)15) + red * sizeof(int) + c2v;
}

template <bool ES, bool CL>
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
  // raw buffer over this workgroup's slab (gfx9 descriptor word 3: 32-bit data format); accesses beyond it are dropped by the hardware
  const __amdgpu_buffer_rsrc_t c2v_rsrc = __builtin_amdgcn_make_buffer_rsrc(c2v, 0, (int)((uint32_t)p.cpb * (uint32_t)p.n_edges * Z), 0x00020000);
  // base graph: row starts and edge words come through the scalar cache
  typedef const __attribute__((address_space(4))) int* cint_p;
  const cint_p ec = (cint_p)p.edges, row_start = (cint_p)p.row_start;
  const unsigned short m9 = (unsigned short)p.sf_m9;
  // CL: this workgroup's (one) code word keeps its messages in LDS behind the soft words and the CRC reduction words
  const uint32_t c2v_lds = CL ? (uint32_t)(reinterpret_cast<char*>(lds) - static_cast<char*>(nullptr)) +
                                    (uint32_t)((((size_t)p.cpb * per_cw + 15) & ~(size_t)15) + (p.crc_order ? (p.cpb == 1 ? 16u : (uint32_t)((p.cpb * H + 63) / 64) * 64u + 8u) : 0u) * sizeof(int))
                              : 0u;

  // Code words are handed out by a counter, not in fixed shares: a workgroup takes words blockIdx.x * cpb ... first and asks for more when
  // it is done.  The CUs do not all run at the same speed (with 1280 persistent workgroups and 12.8 words each by a fixed stride the launch
  // took 25.4 ms for 16,384 words, with one workgroup per word -- the dispatcher hands the next word to whoever is free -- 22.1 ms; same
  // code, same bytes, and a fresh slab per word in the persistent version changes nothing: profiles/r02_ldpc_experiments.txt).
  __shared__ int s_next;
  for (int cw0 = blockIdx.x * p.cpb; cw0 < p.n_cw;) {
    const int  cw      = cw0 + cwl;
    const bool present = (cwl < p.cpb) && (cw < p.n_cw);
    const int  cwi     = (present && p.cw_map) ? (int)p.cw_map[cw] : cw; // row of the LLR / message arrays
    bool       active  = present;
    int        it_done = 0;
    __syncthreads(); // the previous code word's message extraction has finished reading the soft words

    // init_ldpc_dec_c (ldpc_dec_c.c:170-188)
    if (active) {
      const int8_t* llr = reinterpret_cast<const int8_t*>(p.llrs) + (size_t)cwi * p.llr_stride;
      soft2[c]     = (uint16_t)(kBias | (kBias << 8));
      soft2[H + c] = (uint16_t)(kBias | (kBias << 8));
      for (int n = 2; n < p.bgN; n++) {
        // values the first variable-to-check pass treats alike are stored alike: |llr| >= 127 is infinity (kept as +-64), anything
        // else enters as clip(llr, +-63) (its check-to-variable messages are still zero then, ldpc_dec_c.c:345-353)
        auto norm = [](int v) { return v >= 127 ? 64 : (v <= -127 ? -64 : (v > 63 ? 63 : (v < -63 ? -63 : v))); };
        const uint32_t lo = (uint32_t)(norm(llr[(n - 2) * Z + c]) + kBias), hi = (uint32_t)(norm(llr[(n - 2) * Z + H + c]) + kBias);
        soft2[n * H + c] = (uint16_t)(lo | (hi << 8));
      }
      for (int e = 0; e < p.n_edges; e++) {
        if (CL) {
          *reinterpret_cast<__attribute__((address_space(3))) uint16_t*>((uintptr_t)(c2v_lds + (uint32_t)e * Z + (uint32_t)c * 2u)) = (uint16_t)(kBias | (kBias << 8));
        } else {
          c2v[(uint32_t)e * H + coff] = (uint16_t)(kBias | (kBias << 8));
        }
      }
    }
    __syncthreads();

    int e0n = row_start[0], e1n = row_start[1];
    for (int it = 0; it < p.max_iter; it++) {
      for (int l = 0; l < p.n_layers; l++) {
        const int e0 = e0n, deg = e1n - e0n;
        {
          const int ln = l + 1 < p.n_layers ? l + 1 : 0; // fetched one row ahead: the edge words' addresses depend on it
          e0n          = row_start[ln];
          e1n          = row_start[ln + 1];
        }

        switch (deg) {
#define LDPC_CASE(D)                                                                                                   \
  case D:                                                                                                              \
    layer_packed<D, !ES, CL>(ec, sbase, c2v_rsrc, coff, e0, c, Z, H, m9, active, c2v_lds);                                  \
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
        uint32_t*      red   = reinterpret_cast<uint32_t*>(lds + (((size_t)p.cpb * per_cw + 15) & ~(size_t)15));
        const uint32_t order = (uint32_t)p.crc_order, mask = order == 32 ? 0xffffffffu : ((1u << order) - 1u), poly = p.crc_poly & mask;
        uint32_t       r     = 0;
        if (active) {
          uint32_t i0  = (uint32_t)c * 2u * (uint32_t)p.bgK;
          uint32_t col = i0 / Z, pos = i0 - col * Z;
          for (int k = 0; k < 2 * p.bgK; k++) {
            const uint32_t half = pos >= H ? 1u : 0u;
            const uint8_t  v    = reinterpret_cast<const uint8_t*>(sbase)[(col * H + pos - half * H) * 2 + half];
            const uint32_t bit  = v < kBias ? 1u : 0u;
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
        m[i] = reinterpret_cast<const uint8_t*>(sbase)[(col * H + pos - half * H) * 2 + half] < kBias;
        pos += H;
        if (pos >= Z) {
          pos -= Z;
          col++;
        }
      }
    }
    // the next share (the barrier at the top of the loop keeps s_next from being overwritten before everybody has read it)
    if (p.work_counter) {
      if (t == 0) {
        s_next = (int)((unsigned)gridDim.x * (unsigned)p.cpb + atomicAdd(p.work_counter, (unsigned)p.cpb));
      }
      __syncthreads();
      cw0 = s_next;
    } else {
      cw0 += (int)gridDim.x * p.cpb;
    }
  }
}

// the packed kernel applies when ...
bool packed_applies(const Params& p)
{
  if (p.dtype != DT_I8 || p.flood || p.iter_msgs || p.soft_out || (p.Z & 1) || p.Z < 8 || p.sf_m9 <= 0) {
    return false;
  }
  if (const int v = knob(KNOB_LDPC_PACKED); v >= 0) { // development knob: 0 = plain kernel
    return v != 0;
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
  if (!p.work_counter) {
    // static shares (no counter): with early stop code words take different times, and workgroups queued behind the resident ones even the load
    // out (BG1 Z = 384, 8192 words at 3 iterations on average: 1280 slots 2.08 ms, 1536 1.95 ms, 2048 1.90 ms); with fixed iterations more,
    // shorter-lived workgroups do the same for CUs of different speed (16,384 words: 1280 workgroups 25.4 ms, 4096 23.7, 16,384 22.1)
    slots = (p.crc_order || 2 * p.max_slots >= 5 * slots) ? p.max_slots : slots;
  }
  slots            = slots > p.max_slots ? p.max_slots : slots;
  if (const int v = knob(KNOB_LDPC_SLOTS); v > 0) { // development knob
    slots = v <= p.max_slots ? v : slots;
  }
  const int groups = (p.n_cw + p.cpb - 1) / p.cpb;
  return groups < slots ? groups : slots;
}

template <bool ES, bool CL>
static hipError_t launch_packed_es(const Params& p, hipStream_t stream)
{
  static bool attr_set[kMaxDevices] = {};
  const int   dev = current_device();
  if (!attr_set[dev < kMaxDevices && dev >= 0 ? dev : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ldpc_packed_kernel<ES, CL>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) {
      return e;
    }
    attr_set[dev < kMaxDevices && dev >= 0 ? dev : 0] = true;
  }
  int threads = p.cpb * (p.Z / 2);
  threads     = ((threads + 63) / 64) * 64;
  hipLaunchKernelGGL((ldpc_packed_kernel<ES, CL>), dim3(grid_slots_packed(p)), dim3(threads), lds_bytes_packed(p), stream, p);
  return hipGetLastError();
}

// messages in LDS: one code word per workgroup, soft words + messages within the LDS of a CU
bool packed_c2v_lds_fits(const Params& p)
{
  Params q  = p;
  q.cpb     = 1;
  q.c2v_lds = 1;
  return lds_bytes_packed(q) <= 156 * 1024;
}

hipError_t launch_packed(const Params& p, hipStream_t stream)
{
  if (lds_bytes_packed(p) > 156 * 1024 || p.cpb * (p.Z / 2) > 768 || (p.c2v_lds && p.cpb != 1)) {
    return hipErrorInvalidValue;
  }
  if (p.c2v_lds) {
    return p.crc_order ? launch_packed_es<true, true>(p, stream) : launch_packed_es<false, true>(p, stream);
  }
  return p.crc_order ? launch_packed_es<true, false>(p, stream) : launch_packed_es<false, false>(p, stream);
}

} // namespace ldpc
} // namespace phyhip
