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

/* Init-time work belongs to init: srsran_sch_init (sch.c:118-197) allocates, creates the turbo coder / decoder objects and calls
 * srsran_rm_turbo_gentables().  With the reference's own rm_turbo.c still in the link (its non-LUT paths serve the sidelink channels) that call
 * reaches the reference's table builder, not the library's init hook of the same name -- so the binding warms the device path here: the definition
 * in the unmodified sch.o is renamed to srsran_sch_init_ref (objcopy --redefine-sym) and this one takes its name.  Afterwards the first transport
 * block of the calling thread costs what a later one costs (profiles/r04_warm_probe.txt); an application with N PHY workers calls
 * srsran_hip_warmup(N) once more after creating them. */
extern int srsran_sch_init_ref(void* q);
extern int srsran_hip_warmup(uint32_t nof_workers);
int        srsran_sch_init(void* q)
{
  const int ret = srsran_sch_init_ref(q);
  if (ret == 0) {
    (void)srsran_hip_warmup(1);
  }
  return ret;
}
