/* tests/ref_link/tb_tx_bind.c -- the reference-side binding of the LTE transmit entry, compiled against the
 * REFERENCE's headers like a file added to lib/src/phy/phch.  sch.c is compiled unmodified; ITS definition of srsran_dlsch_encode2 (sch.c:625-658:
 * segmentation, bits per symbol, then the static encode_tb) is made weak in the object file and this one takes its place -- srsran_dlsch_encode
 * (:620) and through it srsran_pmch_encode reach it too.  A transport block that srsran_pdsch_encode hands over is encoded as ONE device call
 * (CRC24A, segmentation, CRC24B, turbo coding, rate matching) instead of two per code block. */
#include <stdint.h>

#include "srsran/phy/phch/sch.h"

/* include/srsran_amd/phy_sch_abi.h (declared by hand: its type names are the reference's) */
extern int srsran_hip_encode_tb(srsran_softbuffer_tx_t* softbuffer,
                                srsran_cbsegm_t*        cb_segm,
                                uint32_t                Qm,
                                uint32_t                rv,
                                uint32_t                nof_e_bits,
                                uint8_t*                data,
                                uint8_t*                e_bits);

int srsran_dlsch_encode2(srsran_sch_t* q, srsran_pdsch_cfg_t* cfg, uint8_t* data, uint8_t* e_bits, int tb_idx, uint32_t nof_layers)
{
  (void)q;
  const uint32_t  Nl = (nof_layers != cfg->grant.nof_tb) ? 2 : 1;
  srsran_cbsegm_t cb_segm;
  if (srsran_cbsegm(&cb_segm, (uint32_t)cfg->grant.tb[tb_idx].tbs)) {
    return SRSRAN_ERROR;
  }
  return srsran_hip_encode_tb(cfg->softbuffers.tx[tb_idx],
                              &cb_segm,
                              srsran_mod_bits_x_symbol(cfg->grant.tb[tb_idx].mod) * Nl,
                              (uint32_t)cfg->grant.tb[tb_idx].rv,
                              cfg->grant.tb[tb_idx].nof_bits,
                              data,
                              e_bits);
}
