// nr_sch_device.h -- parameter blocks and launchers of nr_sch_kernels.hip: NR LDPC rate matching (both directions) and
// the LDPC encoder (lib/src/phy/fec/ldpc/ldpc_rm.c, ldpc_encoder.c, ldpc_enc_c.c)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace nrsch {

enum { T_I8 = 0, T_I16 = 1, T_F32 = 2 };

struct CbJob { // one code block
  uint32_t in_off;  // elements
  uint32_t out_off; // elements
  uint32_t E;       // rate-matched length
  uint32_t aux;     // encoder: cdwd_rm_length after clamping / rounding
};

struct RmParams { // init_rm, ldpc_rm.c:113-167
  const void*  in;
  void*        out;
  const CbJob* jobs;
  uint32_t     n_cb;
  uint32_t     Ncb, k0, ini_ex, end_ex; // circular buffer, start, filler positions [ini_ex, end_ex)
  uint32_t     Qm;
  int          type;
};

hipError_t launch_rm_rx(const RmParams& p, uint32_t max_E, hipStream_t stream); // max_E: the largest job.E of the batch
hipError_t launch_rm_tx(const RmParams& p, hipStream_t stream); // uint8 code words -> uint8 rate-matched bits

#define NRSCH_MAX_CORE_TERMS 4
struct EncStep { // one of the three core parity blocks solved after p0: p[unk] = rot(lam[row] + sum rot(p[blk], sh), -ush)
  int row, unk, ush;
  int n_terms;
  int blk[NRSCH_MAX_CORE_TERMS], sh[NRSCH_MAX_CORE_TERMS];
};

struct EncParams {
  const uint8_t* in;  // n_cb messages: bgK * Z bytes (bit per byte, 254 = filler)
  uint8_t*       out; // n_cb code words: (bgN - 2) * Z bytes
  const CbJob*   jobs;
  uint32_t       n_cb;
  const int*     row_start; // bgM + 1
  const int*     edges;     // col | shift << 8
  int            Z, bgN, bgM, bgK;
  int            a;  // p0 = rot(lam0 + lam1 + lam2 + lam3, -a)
  EncStep        step[3];
};

hipError_t launch_encode(const EncParams& p, hipStream_t stream);

} // namespace nrsch
} // namespace phyhip
