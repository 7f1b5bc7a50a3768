// rm_device.h -- kernel parameter blocks and launchers of rm_kernels.hip (turbo rate de-matching, CRC)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace rm {

// one rate-matched code block: out[table[k]] += sum_m in[k + m * out_len]  (k + m * out_len < in_len)
struct RxJob {
  uint32_t in_offset;  // first soft bit of this code block in the input stream (elements)
  uint32_t in_len;     // number of received soft bits E
  uint32_t out_offset; // start of this code block's soft buffer (elements)
  uint32_t out_len;    // 3K + 12
  uint32_t table;      // offset of the position table of (K, rv, layout) in the table pool (uint16 elements)
  uint32_t fresh;      // 1: the soft buffer holds nothing yet (new data): overwrite all of it instead of accumulating
};

struct TbCrcJob {
  uint32_t data_offset; // first byte of the transport block
  uint32_t tbs;         // payload bits (multiple of 8); the 3 parity bytes follow
  // the block's code blocks in THIS launch: two runs of consecutive entries of the decoder's per-block verdicts (blocks of K1, blocks of K2);
  // the parity is only computed when all of them are good (`need` = their number; blocks decoded in earlier HARQ rounds are not listed)
  uint32_t run_start[2], run_len[2], need;
  uint32_t mult; // which 256-entry row of the multiplier table belongs to this tbs: x^(8 * bytes behind lane l's chunk) mod g
};
struct TbCrcResult {
  uint32_t par_rx;   // CRC24A of the payload
  uint32_t par_tx;   // the received parity bytes
  uint32_t computed; // 0: some code block of the transport block failed, no parity taken (sch.c:473-477)
};
// d_cb_ok: the decoder's verdict per code block of the launch (1 = CRC good)
// d_mult: rows of 256 multipliers (tb_crc_multipliers(), one row per distinct tbs of the launch)
hipError_t launch_tb_crc(const uint8_t* d_data, const TbCrcJob* d_jobs, int n_jobs, uint32_t poly, const uint8_t* d_cb_ok, const uint32_t* d_mult,
                         TbCrcResult* d_res, hipStream_t stream, uint8_t* host_data = nullptr);
// the kernel's split of a block of tbs / 8 bytes over its 256 lanes, and what lane l multiplies its chunk's remainder with
void tb_crc_multipliers(uint32_t tbs, uint32_t poly, uint32_t out[256]);

// d_jobs: device array of n_jobs descriptors.  elem8: int8 soft bits (wrapping), else int16.
hipError_t launch_rx(const void* d_in, void* d_out, const uint16_t* d_tables, const RxJob* d_jobs, int n_jobs, bool elem8,
                     hipStream_t stream);

// output-driven form: d_inverse[j] = index of the transmitted bit that lands on soft-buffer position j (0xffff: none),
// out_span = soft-buffer positions covered (3K+12 natural, 3(K+32)+12 decoder layout); jobs as above (d_jobs or uni)
hipError_t launch_rx_gather(const void* d_in, void* d_out, const uint16_t* d_inverse, uint32_t out_span, const RxJob* d_jobs,
                            const RxJob& uni, uint32_t in_stride, uint32_t out_stride, int n_jobs, bool elem8, hipStream_t stream,
                            uint32_t max_in_len = 0); // max_in_len: largest in_len of the batch (0: unknown); small inputs are staged in LDS

// uniform batch: job b = `first` with offsets advanced by b * (in_stride, out_stride); no descriptor array
hipError_t launch_rx_uniform(const void* d_in, void* d_out, const uint16_t* d_table, const RxJob& first, uint32_t in_stride,
                             uint32_t out_stride, int n_jobs, bool elem8, hipStream_t stream);

} // namespace rm
} // namespace phyhip
