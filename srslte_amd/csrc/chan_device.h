// chan_device.h -- small kernels of the grant-level entry points (chan_host.cpp) that have no home in a stage's own file
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace phyhip {
namespace chan {

// UL-SCH channel interleaver on byte-packed bits (TS 36.212 5.2.2.8 without RI bits; sch.c:934-990 ulsch_interleave): the g bits are nof_sym groups of
// Qm bits written row by row into a matrix of `cols` columns (rows = nof_sym / cols) and read column by column: group (c rows + r) of q = group
// (r cols + c) of g.  q_bits: every byte of [0, ceil(nof_sym Qm / 8)) is written.
hipError_t launch_ul_interleave_bits(const uint8_t* g_bits, uint8_t* q_bits, uint32_t nof_sym, uint32_t Qm, uint32_t cols, hipStream_t stream);

} // namespace chan
} // namespace phyhip
