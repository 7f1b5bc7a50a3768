// ldpc_device.h -- kernel parameter block and launcher of ldpc_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace ldpc {

enum { DT_I8 = 0, DT_I16 = 1, DT_F32 = 2 }; // LLR / message type: ldpc_dec_c.c, ldpc_dec_s.c, ldpc_dec_f.c

struct Params {
  const void*     llrs;      // n_cw x (N-2Z) LLRs of type dtype, llr_stride ELEMENTS apart
  uint8_t*        msg;       // n_cw x K*Z bytes (bit per byte), msg_stride apart
  uint8_t*        iter_msgs; // optional: n_cw x max_iter x ceil(K*Z/8) packed hard decisions per iteration
  const int*      row_start; // bgM+1 : first edge of every base-graph row
  const int*      edges;     // per edge: (variable node * Z) | (circular shift V mod Z) << 16
  uint32_t        llr_stride;
  uint32_t        msg_stride;
  int             Z;
  int             bgN;
  int             bgK;
  int             n_layers;
  int             n_edges; // edges of the first n_layers rows
  int             max_iter;
  int             sf; // (int)(scaling_fctr * 100)
  int             n_cw;
  int             cpb; // code words per workgroup
  int             max_slots; // c2v slabs behind c2v_ws: upper bound of the grid
  int             dtype;
  float           sf_f;     // scaling factor of the float decoder
  void*           c2v_ws;   // max_slots x cpb x n_edges x Z check-to-variable messages: one slab per workgroup of the launch
  unsigned int*   work_counter; // packed kernel: zeroed before the launch; workgroups take their next code words from it (nullptr: static shares)
  void*           soft_out; // optional: n_cw x bgN*Z a-posteriori soft bits (parity aid)
  // flooded schedule (int8 only): edges of every variable node in row order over ALL rows; word = edge index | shift << 16
  int             flood;
  int             n_col_edges;
  const int*      col_start; // bgN + 1
  const int*      col_edges;
  // CRC early stop (decode_crc_c): generator without / with its top bit, order (0: off), x^((Z-1-c) bgK) mod g per lane,
  // iterations per code word out (0 = no match)
  uint32_t        crc_poly;
  int             crc_order;
  const uint32_t* crc_mult;
  int*            n_iter_out;
  // two lifted positions per lane (ldpc_packed_kernels.hip): set by the host when packed_applies(); sf_m9 = M with
  // (x * M) >> 9 == x * sf / 100 for every x in 0..127, or 0 when there is no such M
  int             packed;
  int             sf_m9;
  // packed kernel, small batches: the check-to-variable messages of a workgroup's code word live in LDS next to its soft words (one word per
  // workgroup; n_edges * Z more bytes) -- a row then waits for LDS, not for the L2 / HBM round trip of its message loads
  int             c2v_lds;
  // optional indirection: code word i of the batch reads its LLRs from / writes its message to row cw_map[i] (the NR
  // transport-block loop decodes the not-yet-decoded code blocks of a soft buffer in place); n_iter_out stays indexed by i
  const uint32_t* cw_map;
};

#define LDPC_MAX_SLOTS 4096 // resident workgroup slots (256 CUs x up to 4); each owns one c2v slab per code word it holds
int        grid_slots(const Params& p);
hipError_t launch(const Params& p, hipStream_t stream);
size_t     lds_bytes(const Params& p);
bool       packed_applies(const Params& p); // int8, layered, even lifting size, no per-iteration / soft-bit side outputs
hipError_t launch_packed(const Params& p, hipStream_t stream);
bool       packed_c2v_lds_fits(const Params& p); // one code word per workgroup with its messages in LDS fits a CU's LDS

} // namespace ldpc
} // namespace phyhip
