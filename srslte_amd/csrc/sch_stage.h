// sch_stage.h -- the calling thread's transport-block stage (sch_host.cpp) as the grant-level entry points (chan_host.cpp) use it
#pragma once
#include "srsran_amd/phy_sch_abi.h"

#include <functional>
#include <hip/hip_runtime.h>

namespace phyhip {
namespace sch {

// enqueues, on the stage's stream, the kernels that leave the transport block's e bits (int16, or int8 when q->llr_is_8bit) at `d_e_bits`
using FrontEnd = std::function<bool(hipStream_t stream, void* d_e_bits)>;

// one transport block of a call: the arguments of decode_tb_cb (sch.c:370), `front` in place of host e bits when they are made on the device
struct TbItem {
  void*                   q; // srsran_sch_t* (its head: srsran_hip_sch_head_t)
  srsran_softbuffer_rx_t* sb;
  srsran_cbsegm_t*        seg;
  uint32_t                Qm, rv, nof_e_bits;
  const void*             e_bits;
  const FrontEnd*         front;
  uint8_t*                data;
  bool                    ok; // out: what decode_tb_cb returns
};

// the same for several transport blocks of a call at once: `which` are indices into the item list given to decode_tbs_staged, d_e_bits[k] is where block
// which[k]'s e bits are expected.  Called once per run of blocks that share a launch (normally once); the items' own `front` must be set (it marks the
// e bits as device-made) but is not called when a group front end is given.
using GroupFrontEnd = std::function<bool(hipStream_t stream, const uint32_t* which, void* const* d_e_bits, uint32_t n)>;

hipStream_t stage_stream();
void decode_tbs_staged(TbItem* items, uint32_t n, const GroupFrontEnd* group = nullptr);
bool decode_tb_staged(void* q, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, const void* e_bits,
                      const FrontEnd* front, uint8_t* data);

// transmit side: encode_tb on the thread's stage (tcod_host.cpp).  `back` consumes the e bits (byte packed) on the device instead of a download.
using BackEnd = std::function<bool(hipStream_t stream, const uint8_t* d_e_bits)>;
// the same for n transport blocks in one call: ONE coding launch over the code blocks of all of them; `back` gets the device image of the e bits and every
// block's first BYTE in it (a block's bits start on a 256-byte boundary)
struct TxItem {
  srsran_softbuffer_tx_t* sb;
  srsran_cbsegm_t*        seg;
  uint32_t                Qm, rv, nof_e_bits;
  uint8_t*                data; // payload, or NULL: retransmission of what the soft buffer holds
  uint32_t                e_byte_off; // out
};
using GroupBackEnd = std::function<bool(hipStream_t stream, const uint8_t* d_e_bits, const uint32_t* e_byte_off, uint32_t n)>;
int encode_tbs_staged(TxItem* items, uint32_t n, const GroupBackEnd* back);
int encode_tb_staged(srsran_softbuffer_tx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, uint8_t* data, uint8_t* e_bits,
                     const BackEnd* back);

} // namespace sch
} // namespace phyhip
