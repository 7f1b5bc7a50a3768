/*
 * phy_batch.h -- batched, device-resident extension of the drop-in ABI (libsrsran_phy_hip.so).
 *
 * One subframe / one code block per call cannot fill 256 CUs.  These entry points run the same
 * kernels as the handle API of phy_abi.h over a batch of independent units whose buffers already
 * live in HBM (plain device pointers; no torch types).  Semantics per unit are exactly those of
 * the cited reference function.  `stream` is a hipStream_t passed as void* (NULL = default stream);
 * calls are asynchronous on that stream.
 */
#ifndef SRSRAN_AMD_PHY_BATCH_H
#define SRSRAN_AMD_PHY_BATCH_H

#include "srsran_amd/phy_abi.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device plumbing for C hosts (thin wrappers over hipMalloc/hipMemcpy/hipStreamSynchronize) ---- */
SRSRAN_API int         srsran_hip_device_count(void);
SRSRAN_API int         srsran_hip_set_device(int device);
/* N worker threads of ONE process on N GPUs (the reference runs its PHY workers as threads: lib/include/srsran/common/thread_pool.h:48):
 * srsran_hip_set_thread_device(d) binds the CALLING thread to device d from now on, whatever srsran_hip_set_device named as the process
 * default (which stays the device of every thread that has not bound itself); -1 returns the thread to the default.  Every handle, batch
 * object and staging context records the device it was created on; an entry point called from a thread bound to another device returns an
 * error (message through srsran_hip_last_error and on stderr) and launches nothing. */
SRSRAN_API int         srsran_hip_set_thread_device(int device);
SRSRAN_API int         srsran_hip_get_thread_device(void);
/* Worker threads overlap their launches only when their streams sit on different hardware queues: the library puts GPU_MAX_HW_QUEUES=8 into the
 * environment from a constructor unless a value is already there -- which the runtime reads at ITS initialisation.  0: the runtime had been initialised
 * before the library was loaded (a note is printed once): set the variable in the environment instead.  1 otherwise. */
SRSRAN_API int         srsran_hip_hw_queues_requested_in_time(void);
SRSRAN_API void*       srsran_hip_malloc(size_t bytes);
SRSRAN_API void        srsran_hip_free(void* dptr);
SRSRAN_API int         srsran_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
SRSRAN_API int         srsran_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
SRSRAN_API int         srsran_hip_memset(void* dst, int value, size_t bytes, void* stream);
SRSRAN_API int         srsran_hip_stream_sync(void* stream);
SRSRAN_API const char* srsran_hip_last_error(void);
SRSRAN_API const char* srsran_hip_build_info(void);

/* ---- submission queues under the handle API.  The reference runs one worker thread per in-flight subframe, each with its own
 * handles (srsenb/src/phy/lte/cc_worker.cc:212-231, lib/include/srsran/common/thread_pool.h:48).  Calls of srsran_tdec_run_all{,_8bit}
 * and srsran_ldpc_decoder_decode_{c,s,f,crc_c} that are in flight at the same time with the same kernel configuration are merged into
 * batch launches once more callers are in flight than the queue has lanes (group commit: no timer, a lone caller runs at once);
 * results are unchanged.  OFDM subframes always run on the handle's private stream (measured faster: profiles/r02_bench_handle.json).  Default on; SRSRAN_HIP_COALESCE=0 in the environment or srsran_hip_set_coalescing(0) gives every handle its
 * private stream again.  srsran_hip_coalesce_stats: batches launched and calls carried so far (either pointer may be NULL). */
SRSRAN_API void srsran_hip_set_coalescing(int enable);
SRSRAN_API void srsran_hip_coalesce_stats(uint64_t* nof_batches, uint64_t* nof_units);
/* submission queues alive right now: one per kernel shape, at most 12 -- the least recently used idle one is released (streams, pinned
 * staging, batch engine) when a new shape arrives, so a long-running process that walks through many block / lifting sizes stays bounded */
SRSRAN_API uint32_t srsran_hip_coalesce_shapes(void);
/* development knobs (measured kernel alternatives kept behind SRSRAN_HIP_TDEC_VARIANT / SRSRAN_HIP_PSS_VARIANT and a few sizing overrides):
 * their environment variables are read once; this overrides one at run time (value NULL = unset).  Never needed by an application. */
SRSRAN_API int      srsran_hip_dev_knob(const char* env_name, const char* value);

/* Batch objects (every srsran_hip_*_batch_t, srsran_hip_sch_t, srsran_hip_sch_nr_t, srsran_hip_cellsearch_t ...) own device workspace
 * (decoder state, message slabs, correlation buffers): ONE stream per object at a time.  Calls on the same object are ordered by that
 * stream; to run two batches concurrently create two objects. */
/* ---- turbo decoder: srsran_tdec_run_all (turbodecoder.c:536-549) over n_cb code blocks ---- */
typedef struct srsran_hip_tdec_batch srsran_hip_tdec_batch_t;

/* impl: SRSRAN_TDEC_AUTO / _GENERIC / _SSE_WINDOW / _AVX_WINDOW / _SSE8_WINDOW / _AVX8_WINDOW select which
 * reference decoder is reproduced bit-exactly (AUTO = the reference's choice on an AVX2 host for the 16-bit
 * API, turbodecoder.c:381-408).  create_8bit: AUTO resolves as srsran_tdec_run_all_8bit does (32 sub-blocks
 * K%32==0 && K>2048, 16 sub-blocks K%16==0 && K>800, else the 16-bit decoders on widened LLRs, :410-478). */
SRSRAN_API int  srsran_hip_tdec_batch_create(srsran_hip_tdec_batch_t** h, uint32_t long_cb, uint32_t max_nof_cb, int impl);
SRSRAN_API int  srsran_hip_tdec_batch_create_8bit(srsran_hip_tdec_batch_t** h, uint32_t long_cb, uint32_t max_nof_cb, int impl);
SRSRAN_API void srsran_hip_tdec_batch_free(srsran_hip_tdec_batch_t* h);
/* d_input : n_cb blocks of int16 LLRs, `in_stride` int16 apart.  sb_layout = 0: natural order
 *           [s p0 p1]xK + 12 tail (3K+12);  sb_layout = 1: srsran_rm_turbo_rx_lut sub-block layout
 *           (3(K+32)+12, turbodecoder_iter.h:88-102; window impls only).
 * d_output: n_cb blocks of K/8 bytes (MSB first), `out_stride` bytes apart. */
SRSRAN_API int  srsran_hip_tdec_batch_run(srsran_hip_tdec_batch_t* h, const int16_t* d_input, uint32_t in_stride,
                                          uint8_t* d_output, uint32_t out_stride, uint32_t n_cb,
                                          uint32_t nof_iterations, int sb_layout, void* stream);
/* same with int8 LLRs (srsran_tdec_run_all_8bit, turbodecoder.c:560-577); in_stride in int8 elements.  Either
 * entry point works on either kind of batch object: the LLRs are converted as the reference does
 * (convert_8_to_16 / convert_16_to_8, turbodecoder.c:443-453). */
SRSRAN_API int  srsran_hip_tdec_batch_run_8bit(srsran_hip_tdec_batch_t* h, const int8_t* d_input, uint32_t in_stride,
                                               uint8_t* d_output, uint32_t out_stride, uint32_t n_cb,
                                               uint32_t nof_iterations, int sb_layout, void* stream);
/* debug/parity aid: copy the SISO output of the last half iteration (K int16 per CB, natural order) */
SRSRAN_API int  srsran_hip_tdec_batch_last_llr(srsran_hip_tdec_batch_t* h, int16_t* d_llr, uint32_t n_cb, void* stream);

/* ---- LDPC: srsran_ldpc_decoder_decode_c (ldpc_decoder.c:657-685, int8 layered) over n_cw words ---- */
typedef struct srsran_hip_ldpc_batch srsran_hip_ldpc_batch_t;

/* Device memory of an object: up to 2048 slabs of check-to-variable messages, edges x ls bytes per code word each for the int8 decoders
 * (121 KB for BG1 ls = 384), one per workgroup of a launch. */
SRSRAN_API int  srsran_hip_ldpc_batch_create(srsran_hip_ldpc_batch_t** h, srsran_basegraph_t bg, uint16_t ls,
                                             float scaling_fctr, uint32_t max_nof_iter, uint32_t max_nof_cw);
/* type: SRSRAN_LDPC_DECODER_F (float LLRs, ldpc_dec_f.c), _S (int16, ldpc_dec_s.c) or the int8 family _C / _C_AVX2 /
 * _C_AVX512; all layered min-sum under the schedule of ldpc_decoder.c:44-104.  Use run_typed with LLRs of that type;
 * llr_stride counts LLRs, not bytes. */
SRSRAN_API int  srsran_hip_ldpc_batch_create_typed(srsran_hip_ldpc_batch_t** h, srsran_basegraph_t bg, uint16_t ls, float scaling_fctr,
                                                   uint32_t max_nof_iter, uint32_t max_nof_cw, srsran_ldpc_decoder_type_t type);
SRSRAN_API int  srsran_hip_ldpc_batch_run_typed(srsran_hip_ldpc_batch_t* h, const void* d_llrs, uint32_t llr_stride,
                                                uint8_t* d_message, uint32_t msg_stride, uint32_t n_cw,
                                                uint32_t cdwd_rm_length, uint8_t* d_iter_msgs, void* stream);
/* srsran_ldpc_decoder_decode_crc_c (ldpc_decoder.c:87-99,682-685) for a batch: int8 decoders only (layered or flooded).  Every
 * code word leaves the iteration loop at the first iteration whose hard decisions pass the CRC (generator crc_polynom of
 * crc_order bits over the first liftK - crc_order bits, compared with the last crc_order); d_nof_iterations[i] receives that
 * iteration count, 0 when it never matched (the reference's return value). */
SRSRAN_API int  srsran_hip_ldpc_batch_run_crc(srsran_hip_ldpc_batch_t* h, const int8_t* d_llrs, uint32_t llr_stride, uint8_t* d_message,
                                              uint32_t msg_stride, uint32_t n_cw, uint32_t cdwd_rm_length, uint32_t crc_polynom,
                                              uint32_t crc_order, int32_t* d_nof_iterations, void* stream);
/* The same with an indirection: code word i of the batch lives in row d_cw_map[i] of the LLR / message arrays (strides as above);
 * d_nof_iterations is indexed by i.  What sch_nr_decode (sch_nr.c:567-665) needs to decode the not yet decoded code blocks of a
 * soft buffer in place. */
SRSRAN_API int  srsran_hip_ldpc_batch_run_crc_map(srsran_hip_ldpc_batch_t* h, const int8_t* d_llrs, uint32_t llr_stride, uint8_t* d_message,
                                                  uint32_t msg_stride, const uint32_t* d_cw_map, uint32_t n_cw, uint32_t cdwd_rm_length,
                                                  uint32_t crc_polynom, uint32_t crc_order, int32_t* d_nof_iterations, void* stream);
SRSRAN_API void srsran_hip_ldpc_batch_free(srsran_hip_ldpc_batch_t* h);
/* d_llrs   : n_cw x (N-2Z) int8 (only the first cdwd_rm_length... all N-2Z are read, as the reference does),
 *            `llr_stride` bytes apart;  d_message: n_cw x K bytes, one bit per byte, `msg_stride` apart.
 * d_iter_msgs (optional, may be NULL): n_cw x max_nof_iter x K/8 packed hard decisions after every
 *            iteration (used by the handle API to reproduce the CRC early stop of decode_crc_c). */
SRSRAN_API int  srsran_hip_ldpc_batch_run(srsran_hip_ldpc_batch_t* h, const int8_t* d_llrs, uint32_t llr_stride,
                                          uint8_t* d_message, uint32_t msg_stride, uint32_t n_cw,
                                          uint32_t cdwd_rm_length, uint8_t* d_iter_msgs, void* stream);

/* ---- OFDM: srsran_ofdm_rx_sf / srsran_ofdm_tx_sf (ofdm.c:453-466,562-576) over n_sf subframes ---- */
typedef struct srsran_hip_ofdm_batch srsran_hip_ofdm_batch_t;

/* cfg: as for srsran_ofdm_rx_init_cfg / tx_init_cfg (in_buffer/out_buffer ignored). */
SRSRAN_API int  srsran_hip_ofdm_batch_create(srsran_hip_ofdm_batch_t** h, const srsran_ofdm_cfg_t* cfg, srsran_dft_dir_t dir);
SRSRAN_API void srsran_hip_ofdm_batch_free(srsran_hip_ofdm_batch_t* h);
SRSRAN_API uint32_t srsran_hip_ofdm_batch_sf_sz(srsran_hip_ofdm_batch_t* h);    /* time samples per subframe */
SRSRAN_API uint32_t srsran_hip_ofdm_batch_sf_re(srsran_hip_ofdm_batch_t* h);    /* resource elements per subframe */
/* rx: d_in n_sf x sf_sz time samples -> d_out n_sf x sf_re REs.  The input is NOT modified (the
 * reference multiplies in_buffer by the shift table in place, ofdm.c:455-457; here it is applied on load). */
/* MBSFN objects (cfg.sf_type = SRSRAN_SF_MBSFN, extended CP): srsran_ofdm_set_non_mbsfn_region (ofdm.c:214), default 2 */
SRSRAN_API int  srsran_hip_ofdm_batch_set_non_mbsfn_region(srsran_hip_ofdm_batch_t* h, uint8_t non_mbsfn_region);
SRSRAN_API int  srsran_hip_ofdm_batch_rx(srsran_hip_ofdm_batch_t* h, const cf_t* d_in, cf_t* d_out, uint32_t n_sf, void* stream);
/* tx: d_in n_sf x sf_re REs -> d_out n_sf x sf_sz time samples */
SRSRAN_API int  srsran_hip_ofdm_batch_tx(srsran_hip_ofdm_batch_t* h, const cf_t* d_in, cf_t* d_out, uint32_t n_sf, void* stream);

/* ---- generic DFT / SC-FDMA transform precoding: srsran_dft_run_c (dft_fftw.c:336-354) over `how_many`
 *      back-to-back transforms of `dft_points` (<= 4096) samples, options fused ---- */
typedef struct srsran_hip_dft_batch srsran_hip_dft_batch_t;

SRSRAN_API int  srsran_hip_dft_batch_create(srsran_hip_dft_batch_t** h, int dft_points, srsran_dft_dir_t dir, bool mirror, bool dc, bool norm);
SRSRAN_API void srsran_hip_dft_batch_free(srsran_hip_dft_batch_t* h);
SRSRAN_API int  srsran_hip_dft_batch_run(srsran_hip_dft_batch_t* h, const cf_t* d_in, cf_t* d_out, uint32_t how_many, void* stream);

#ifdef __cplusplus
}
#endif
#endif
