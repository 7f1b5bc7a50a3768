/* tests/ref_link/tb_bind.c -- the reference-side binding of the transport-block seam.
 *
 * The reference's sch.c is compiled UNMODIFIED; its own definition of decode_tb_cb (sch.c:370, the one non-static function between
 * srsran_dlsch_decode2 / srsran_ulsch_decode and the per-code-block rate-matching / turbo / CRC calls) is made a weak symbol in the
 * object file (objcopy --weaken-symbol, tests/ref_link/Makefile), and this definition takes its place at link time: every transport
 * block the reference's PDSCH / PUSCH / PMCH objects decode then goes to the device as ONE call.  A maintainer would delete the
 * function body from sch.c instead and let the library's own `decode_tb_cb` symbol resolve the call. */
#include <stdbool.h>
#include <stdint.h>

#include "srsran_amd/phy_sch_abi.h"

bool decode_tb_cb(void* q, srsran_softbuffer_rx_t* softbuffer, srsran_cbsegm_t* cb_segm, uint32_t Qm, uint32_t rv, uint32_t nof_e_bits, void* e_bits,
                  uint8_t* data)
{
  return srsran_hip_decode_tb_cb(q, softbuffer, cb_segm, Qm, rv, nof_e_bits, e_bits, data);
}
