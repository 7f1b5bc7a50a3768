"""ctypes binding of libsrsran_phy_hip.so -- the thin Python mirror of the reference's C interface.

Nothing is computed here: every call goes through the C ABI declared in include/srsran_amd/*.h.
There is NO fallback: if the shared library (or, at run time, a HIP device) is missing the calls fail
loudly.  The structures mirror lib/include/srsran/phy/{dft/dft.h,dft/ofdm.h,fec/turbo/turbodecoder.h,
fec/ldpc/ldpc_decoder.h} of the reference field by field.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRSRAN_HIP_LIB: another build of the same library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("SRSRAN_HIP_LIB") or os.path.join(_HERE, "lib", "libsrsran_phy_hip.so")

SRSRAN_SUCCESS = 0
SRSRAN_ERROR = -1
SRSRAN_ERROR_INVALID_INPUTS = -2

# srsran_tdec_impl_type_t (turbodecoder_impl.h:28-38)
TDEC_AUTO, TDEC_GENERIC, TDEC_SSE, TDEC_SSE_WINDOW, TDEC_NEON_WINDOW, TDEC_AVX_WINDOW, TDEC_SSE8_WINDOW, TDEC_AVX8_WINDOW = range(8)
# srsran_ldpc_decoder_type_t (ldpc_decoder.h:41-55)
LDPC_F, LDPC_S, LDPC_C, LDPC_C_FLOOD, LDPC_C_AVX2, LDPC_C_AVX2_FLOOD, LDPC_C_AVX512, LDPC_C_AVX512_FLOOD = range(8)
BG1, BG2 = 0, 1
CP_NORM, CP_EXT = 0, 1
SF_NORM, SF_MBSFN = 0, 1
DFT_FORWARD, DFT_BACKWARD = 0, 1


class DftPlan(C.Structure):
    _fields_ = [("init_size", C.c_int), ("size", C.c_int), ("in_", C.c_void_p), ("out", C.c_void_p), ("p", C.c_void_p),
                ("is_guru", C.c_bool), ("forward", C.c_bool), ("mirror", C.c_bool), ("db", C.c_bool), ("norm", C.c_bool),
                ("dc", C.c_bool), ("dir", C.c_int), ("mode", C.c_int)]


class DftPrecoding(C.Structure):
    _fields_ = [("max_prb", C.c_uint32), ("dft_plan", DftPlan * 111)]


class ConvFftCc(C.Structure):
    _fields_ = [("input_fft", C.c_void_p), ("filter_fft", C.c_void_p), ("output_fft", C.c_void_p), ("output_fft2", C.c_void_p),
                ("input_len", C.c_uint32), ("filter_len", C.c_uint32), ("output_len", C.c_uint32), ("max_input_len", C.c_uint32),
                ("max_filter_len", C.c_uint32), ("input_plan", DftPlan), ("filter_plan", DftPlan), ("output_plan", DftPlan)]


class FiltCc(C.Structure):
    _fields_ = [("filter_input", C.c_void_p), ("downsampled_input", C.c_void_p), ("filter_output", C.c_void_p),
                ("is_decimator", C.c_bool), ("factor", C.c_int), ("num_taps", C.c_int), ("taps", C.c_void_p)]


class Pss(C.Structure):
    _fields_ = [("conv_fft", ConvFftCc), ("filter", FiltCc), ("decimate", C.c_int), ("max_frame_size", C.c_uint32),
                ("max_fft_size", C.c_uint32), ("frame_size", C.c_uint32), ("N_id_2", C.c_uint32), ("fft_size", C.c_uint32),
                ("pss_signal_freq_full", C.c_void_p * 3), ("pss_signal_time", C.c_void_p * 3),
                ("pss_signal_time_scale", C.c_void_p * 3), ("pss_signal_freq", (C.c_float * 124) * 3), ("tmp_input", C.c_void_p),
                ("conv_output", C.c_void_p), ("conv_output_abs", C.c_void_p), ("ema_alpha", C.c_float),
                ("conv_output_avg", C.POINTER(C.c_float)), ("peak_value", C.c_float), ("filter_pss_enable", C.c_bool),
                ("dftp_input", DftPlan), ("idftp_input", DftPlan), ("tmp_fft", C.c_float * 4096), ("tmp_fft2", C.c_float * 4096),
                ("tmp_ce", C.c_float * 124), ("chest_on_filter", C.c_bool)]


class SssFcTables(C.Structure):
    _fields_ = [("z1", (C.c_float * 31) * 31), ("c", (C.c_float * 31) * 2), ("s", (C.c_float * 31) * 31), ("sd", (C.c_float * 30) * 31)]


class Sss(C.Structure):
    _fields_ = [("dftp_input", DftPlan), ("fft_size", C.c_uint32), ("max_fft_size", C.c_uint32), ("corr_peak_threshold", C.c_float),
                ("symbol_sz", C.c_uint32), ("subframe_sz", C.c_uint32), ("N_id_2", C.c_uint32), ("N_id_1_table", (C.c_uint32 * 30) * 30),
                ("fc_tables", SssFcTables * 3), ("corr_output_m0", C.c_float * 31), ("corr_output_m1", C.c_float * 31)]


class Cexptab(C.Structure):
    _fields_ = [("size", C.c_uint32), ("tab", C.c_void_p)]


class Cfo(C.Structure):
    _fields_ = [("last_freq", C.c_float), ("tol", C.c_float), ("nsamples", C.c_int), ("max_samples", C.c_int), ("tab", Cexptab),
                ("cur_cexp", C.c_void_p)]


class CpSynch(C.Structure):
    _fields_ = [("corr", C.c_void_p), ("symbol_sz", C.c_uint32), ("max_symbol_sz", C.c_uint32)]


class Sync(C.Structure):
    """srsran_sync_t (sync.h:50-133)"""
    _fields_ = [("pss", Pss), ("pss_i", Pss * 2), ("sss", Sss), ("cp_synch", CpSynch), ("cfo_i_corr", C.c_void_p * 2),
                ("decimate", C.c_int), ("threshold", C.c_float), ("peak_value", C.c_float), ("N_id_2", C.c_uint32),
                ("N_id_1", C.c_uint32), ("sf_idx", C.c_uint32), ("fft_size", C.c_uint32), ("frame_size", C.c_uint32),
                ("max_offset", C.c_uint32), ("nof_symbols", C.c_uint32), ("cp_len", C.c_uint32), ("current_cfo_tol", C.c_float),
                ("sss_alg", C.c_int), ("detect_cp", C.c_bool), ("sss_en", C.c_bool), ("cp", C.c_int), ("m0", C.c_uint32),
                ("m1", C.c_uint32), ("m0_value", C.c_float), ("m1_value", C.c_float), ("M_norm_avg", C.c_float),
                ("M_ext_avg", C.c_float), ("temp", C.c_void_p), ("max_frame_size", C.c_uint32), ("frame_type", C.c_int),
                ("detect_frame_type", C.c_bool), ("cfo_cp_enable", C.c_bool), ("cfo_pss_enable", C.c_bool),
                ("cfo_i_enable", C.c_bool), ("cfo_cp_is_set", C.c_bool), ("cfo_pss_is_set", C.c_bool),
                ("cfo_i_initiated", C.c_bool), ("cfo_cp_mean", C.c_float), ("cfo_pss", C.c_float), ("cfo_pss_mean", C.c_float),
                ("cfo_i_value", C.c_int), ("cfo_ema_alpha", C.c_float), ("cfo_cp_nsymbols", C.c_uint32),
                ("cfo_corr_frame", Cfo), ("cfo_corr_symbol", Cfo), ("sss_channel_equalize", C.c_bool),
                ("pss_filtering_enabled", C.c_bool), ("sss_filt", C.c_float * 4096), ("pss_filt", C.c_float * 4096),
                ("sss_generated", C.c_bool), ("sss_detected", C.c_bool), ("sss_available", C.c_bool), ("sss_corr", C.c_float),
                ("idftp_sss", DftPlan), ("sss_recv", C.c_float * 4096), ("sss_signal", (C.c_float * 4096) * 2)]


SYNC_FOUND, SYNC_FOUND_NOSPACE, SYNC_NOFOUND, SYNC_ERROR = 1, 2, 0, -1
SSS_DIFF, SSS_FULL, SSS_PARTIAL_3 = 0, 1, 2
FDD, TDD = 0, 1


class HipTb(C.Structure):
    _fields_ = [("tbs", C.c_uint32), ("Qm", C.c_uint32), ("rv", C.c_uint32), ("nof_e_bits", C.c_uint32), ("e_offset", C.c_uint32),
                ("data_offset", C.c_uint32), ("first_cb", C.c_uint32)]


class HipTbResult(C.Structure):
    _fields_ = [("crc_ok", C.c_int32), ("avg_iterations", C.c_float), ("nof_cb", C.c_uint32)]


class Cbsegm(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("F", "C", "K1", "K2", "K1_idx", "K2_idx", "C1", "C2", "tbs", "L_tb", "L_cb", "Z")]


SOFTBUFFER_CB_SIZE = 18600


class Tcod(C.Structure):  # srsran_tcod_t, turbocoder.h:46-49
    _fields_ = [("max_long_cb", C.c_uint32), ("temp", C.c_void_p)]


class HipDemodJob(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("mod", "nof_symbols", "symbol_offset", "llr_offset", "seed", "descramble")]


LLR_SHORT, LLR_BYTE, LLR_FLOAT, MOD_NONE = 0, 1, 2, 5


class HipGrantTb(C.Structure):  # srsran_hip_grant_tb_t, include/srsran_amd/phy_chan_abi.h
    _fields_ = [(n, C.c_uint32) for n in ("mod", "tbs", "rv", "nof_re", "seed", "max_nof_iterations", "llr_is_8bit", "nl")]


class HipGrantRes(C.Structure):
    _fields_ = [("crc_ok", C.c_int32), ("avg_iterations_block", C.c_float), ("epre", C.c_float)]


class HipPuschRx(C.Structure):
    _fields_ = [("tb", HipGrantTb), ("cell_nof_prb", C.c_uint32), ("cp_nsymb", C.c_uint32), ("n_prb_tilde", C.c_uint32 * 2), ("L_prb", C.c_uint32),
                ("shortened", C.c_uint32), ("noise_estimate", C.c_float), ("meas_epre", C.c_uint32)]


class HipPdschRx(C.Structure):
    _fields_ = [("tb", HipGrantTb), ("scaling", C.c_float), ("noise_estimate", C.c_float)]


class HipPdschTx(C.Structure):
    _fields_ = [("tb", HipGrantTb), ("scaling", C.c_float)]


class SoftbufferRx(C.Structure):  # srsran_softbuffer_rx_t, softbuffer.h:40-47
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_f", C.POINTER(C.c_void_p)), ("data", C.POINTER(C.c_void_p)),
                ("cb_crc", C.POINTER(C.c_bool)), ("tb_crc", C.c_bool)]


class SoftbufferTx(C.Structure):  # srsran_softbuffer_tx_t, softbuffer.h:49-53
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_b", C.POINTER(C.c_void_p))]


class LdpcRm(C.Structure):  # srsran_ldpc_rm_t, ldpc_rm.h:37-52
    _fields_ = [("ptr", C.c_void_p), ("bg", C.c_int), ("ls", C.c_uint16), ("N", C.c_uint32), ("E", C.c_uint32), ("K", C.c_uint32),
                ("F", C.c_uint32), ("k0", C.c_uint32), ("mod_order", C.c_uint32), ("Ncb", C.c_uint32)]


class LdpcEncoder(C.Structure):  # srsran_ldpc_encoder_t, ldpc_encoder.h:53-74
    _fields_ = [("ptr", C.c_void_p), ("bg", C.c_int), ("ls", C.c_uint16), ("bgN", C.c_uint8), ("liftN", C.c_uint16), ("bgM", C.c_uint8),
                ("liftM", C.c_uint16), ("bgK", C.c_uint8), ("liftK", C.c_uint16), ("pcm", C.c_void_p), ("free", C.c_void_p),
                ("encode", C.c_void_p), ("encode_high_rate", C.c_void_p), ("encode_high_rate_avx2", C.c_void_p),
                ("encode_high_rate_avx512", C.c_void_p)]


class HipLdpcCb(C.Structure):
    _fields_ = [("in_offset", C.c_uint32), ("out_offset", C.c_uint32), ("E", C.c_uint32)]


class HipNrTb(C.Structure):  # srsran_hip_nr_tb_t
    _fields_ = [("R", C.c_double)] + [(n, C.c_uint32) for n in ("tbs", "mod", "rv", "N_L", "nof_bits", "Nref", "e_offset", "payload_offset",
                                                                 "first_cb", "reserved")]


class HipNrTbResult(C.Structure):
    _fields_ = [("crc_ok", C.c_int32), ("all_decoded", C.c_int32), ("avg_iter", C.c_float), ("nof_cb", C.c_uint32)]


class HipCell(C.Structure):
    _fields_ = [("peak_pos", C.c_int32), ("peak_value", C.c_float), ("psr", C.c_float), ("sss_available", C.c_int32),
                ("m0", C.c_uint32), ("m1", C.c_uint32), ("m0_value", C.c_float), ("m1_value", C.c_float), ("N_id_1", C.c_int32),
                ("sf_idx", C.c_uint32)]


class OfdmCfg(C.Structure):
    _fields_ = [("nof_prb", C.c_uint32), ("in_buffer", C.c_void_p), ("out_buffer", C.c_void_p), ("cp", C.c_int),
                ("sf_type", C.c_int), ("normalize", C.c_bool), ("freq_shift_f", C.c_float),
                ("rx_window_offset", C.c_float), ("symbol_sz", C.c_uint32), ("keep_dc", C.c_bool)]


class Ofdm(C.Structure):
    _fields_ = [("cfg", OfdmCfg), ("fft_plan", DftPlan), ("fft_plan_sf", DftPlan * 2), ("max_prb", C.c_uint32),
                ("nof_symbols", C.c_uint32), ("nof_guards", C.c_uint32), ("nof_re", C.c_uint32), ("slot_sz", C.c_uint32),
                ("sf_sz", C.c_uint32), ("tmp", C.c_void_p), ("mbsfn_subframe", C.c_bool), ("mbsfn_guard_len", C.c_uint32),
                ("nof_symbols_mbsfn", C.c_uint32), ("non_mbsfn_region", C.c_uint8), ("window_offset_n", C.c_uint32),
                ("shift_buffer", C.c_void_p), ("window_offset_buffer", C.c_void_p)]


class TcInterl(C.Structure):
    _fields_ = [("forward", C.POINTER(C.c_uint16)), ("reverse", C.POINTER(C.c_uint16)), ("max_long_cb", C.c_uint32)]


class Tdec(C.Structure):
    _fields_ = [("max_long_cb", C.c_uint32), ("dec8_hdlr", C.c_void_p * 2), ("dec16_hdlr", C.c_void_p * 3),
                ("dec8", C.c_void_p * 2), ("dec16", C.c_void_p * 3), ("nof_blocks8", C.c_int * 2),
                ("nof_blocks16", C.c_int * 3), ("app1", C.c_void_p), ("app2", C.c_void_p), ("ext1", C.c_void_p),
                ("ext2", C.c_void_p), ("syst0", C.c_void_p), ("parity0", C.c_void_p), ("parity1", C.c_void_p),
                ("input_conv", C.c_void_p), ("force_not_sb", C.c_bool), ("dec_type", C.c_int),
                ("current_llr_type", C.c_int), ("current_dec", C.c_uint32), ("current_long_cb", C.c_uint32),
                ("current_inter_idx", C.c_uint32), ("current_cbidx", C.c_int), ("interleaver", (TcInterl * 188) * 4),
                ("n_iter", C.c_int)]


class Crc(C.Structure):
    _fields_ = [("table", C.c_uint64 * 256), ("polynom", C.c_int), ("order", C.c_int), ("crcinit", C.c_uint64),
                ("crcmask", C.c_uint64), ("crchighbit", C.c_uint64), ("srsran_crc_out", C.c_uint32)]


class LdpcDecoderArgs(C.Structure):
    _fields_ = [("type", C.c_int), ("bg", C.c_int), ("ls", C.c_uint16), ("scaling_fctr", C.c_float),
                ("max_nof_iter", C.c_uint32)]


class LdpcDecoder(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bg", C.c_int), ("ls", C.c_uint16), ("max_nof_iter", C.c_uint32),
                ("bgN", C.c_uint8), ("liftN", C.c_uint16), ("bgM", C.c_uint8), ("liftM", C.c_uint16), ("bgK", C.c_uint8),
                ("liftK", C.c_uint16), ("pcm", C.c_void_p), ("var_indices", C.c_void_p), ("scaling_fctr", C.c_float),
                ("free", C.c_void_p), ("decode_f", C.c_void_p), ("decode_s", C.c_void_p), ("decode_c", C.c_void_p)]


_lib = None


def lib():
    """Load libsrsran_phy_hip.so (built by srslte_amd.build).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: run `python -m srslte_amd.build` (hipcc, gfx950). "
                               "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u32, i32, u8p = C.c_void_p, C.c_uint32, C.c_int, C.c_void_p
        sig = {
            "srsran_hip_device_count": (i32, []),
            "srsran_hip_set_device": (i32, [i32]),
            "srsran_hip_set_thread_device": (i32, [i32]),
            "srsran_hip_get_thread_device": (i32, []),
            "srsran_hip_warmup": (i32, [u32]),
            "srsran_hip_malloc": (vp, [C.c_size_t]),
            "srsran_hip_free": (None, [vp]),
            "srsran_hip_memcpy_h2d": (i32, [vp, vp, C.c_size_t, vp]),
            "srsran_hip_memcpy_d2h": (i32, [vp, vp, C.c_size_t, vp]),
            "srsran_hip_memset": (i32, [vp, i32, C.c_size_t, vp]),
            "srsran_hip_stream_sync": (i32, [vp]),
            "srsran_hip_last_error": (C.c_char_p, []),
            "srsran_hip_build_info": (C.c_char_p, []),
            "srsran_hip_set_coalescing": (None, [i32]),
            "srsran_hip_coalesce_stats": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "srsran_hip_coalesce_shapes": (C.c_uint32, []),
            "srsran_hip_dev_knob": (C.c_int, [C.c_char_p, C.c_char_p]),
            "srsran_hip_tdec_batch_create": (i32, [C.POINTER(vp), u32, u32, i32]),
            "srsran_hip_tdec_batch_create_8bit": (i32, [C.POINTER(vp), u32, u32, i32]),
            "srsran_hip_tdec_batch_free": (None, [vp]),
            "srsran_hip_tdec_batch_run_8bit": (i32, [vp, vp, u32, vp, u32, u32, u32, i32, vp]),
            "srsran_hip_tdec_batch_run_dbg_8bit": (i32, [vp, vp, u32, vp, u32, u32, u32, u32, i32, vp]),
            "srsran_hip_tdec_batch_run": (i32, [vp, vp, u32, vp, u32, u32, u32, i32, vp]),
            "srsran_hip_tdec_batch_run_dbg": (i32, [vp, vp, u32, vp, u32, u32, u32, u32, i32, vp]),
            "srsran_hip_tdec_batch_last_llr": (i32, [vp, vp, u32, vp]),
            "srsran_hip_ldpc_batch_create": (i32, [C.POINTER(vp), i32, C.c_uint16, C.c_float, u32, u32]),
            "srsran_hip_ldpc_batch_create_typed": (i32, [C.POINTER(vp), i32, C.c_uint16, C.c_float, u32, u32, i32]),
            "srsran_hip_ldpc_batch_run_typed": (i32, [vp, vp, u32, vp, u32, u32, u32, vp, vp]),
            "srsran_hip_ldpc_batch_run_crc": (i32, [vp, vp, u32, vp, u32, u32, u32, u32, u32, vp, vp]),
            "srsran_hip_ldpc_batch_run_crc_map": (i32, [vp, vp, u32, vp, u32, vp, u32, u32, u32, u32, vp, vp]),
            "srsran_hip_sch_nr_create": (i32, [C.POINTER(vp), C.c_float, u32, u32]),
            "srsran_hip_sch_nr_free": (None, [vp]),
            "srsran_hip_sch_nr_encode": (i32, [vp, vp, C.POINTER(HipNrTb), u32, vp, vp]),
            "srsran_hip_sch_nr_decode": (i32, [vp, vp, C.POINTER(HipNrTb), u32, vp, u32, vp, vp, u32, vp, C.POINTER(HipNrTbResult), vp]),
            "srsran_hip_ldpc_batch_run_dbg": (i32, [vp, vp, u32, vp, u32, u32, u32, vp, vp]),
            "srsran_hip_ldpc_batch_free": (None, [vp]),
            "srsran_hip_ldpc_batch_run": (i32, [vp, vp, u32, vp, u32, u32, u32, vp, vp]),
            "srsran_hip_ofdm_batch_create": (i32, [C.POINTER(vp), C.POINTER(OfdmCfg), i32]),
            "srsran_hip_ofdm_batch_free": (None, [vp]),
            "srsran_hip_ofdm_batch_sf_sz": (u32, [vp]),
            "srsran_hip_ofdm_batch_sf_re": (u32, [vp]),
            "srsran_hip_ofdm_batch_set_non_mbsfn_region": (i32, [vp, C.c_uint8]),
            "srsran_hip_ofdm_batch_rx": (i32, [vp, vp, vp, u32, vp]),
            "srsran_hip_ofdm_batch_tx": (i32, [vp, vp, vp, u32, vp]),
            "srsran_tdec_init": (i32, [C.POINTER(Tdec), u32]),
            "srsran_tdec_init_manual": (i32, [C.POINTER(Tdec), u32, i32]),
            "srsran_tdec_free": (None, [C.POINTER(Tdec)]),
            "srsran_tdec_force_not_sb": (None, [C.POINTER(Tdec)]),
            "srsran_tdec_new_cb": (i32, [C.POINTER(Tdec), u32]),
            "srsran_tdec_get_nof_iterations": (i32, [C.POINTER(Tdec)]),
            "srsran_tdec_autoimp_get_subblocks": (u32, [u32]),
            "srsran_tdec_autoimp_get_subblocks_8bit": (u32, [u32]),
            "srsran_tdec_iteration": (None, [C.POINTER(Tdec), vp, vp]),
            "srsran_tdec_run_all": (i32, [C.POINTER(Tdec), vp, vp, u32, u32]),
            "srsran_tdec_iteration_8bit": (None, [C.POINTER(Tdec), vp, vp]),
            "srsran_tdec_run_all_8bit": (i32, [C.POINTER(Tdec), vp, vp, u32, u32]),
            "srsran_tc_interl_init": (i32, [C.POINTER(TcInterl), u32]),
            "srsran_tc_interl_free": (None, [C.POINTER(TcInterl)]),
            "srsran_tc_interl_LTE_gen": (i32, [C.POINTER(TcInterl), u32]),
            "srsran_tc_interl_LTE_gen_interl": (i32, [C.POINTER(TcInterl), u32, u32]),
            "srsran_cbsegm_cbindex": (i32, [u32]),
            "srsran_cbsegm_cbsize": (i32, [u32]),
            "srsran_ldpc_decoder_init": (i32, [C.POINTER(LdpcDecoder), C.POINTER(LdpcDecoderArgs)]),
            "srsran_ldpc_decoder_free": (None, [C.POINTER(LdpcDecoder)]),
            "srsran_ldpc_decoder_decode_c": (i32, [C.POINTER(LdpcDecoder), vp, vp, u32]),
            "srsran_ldpc_decoder_decode_f": (i32, [C.POINTER(LdpcDecoder), vp, vp, u32]),
            "srsran_ldpc_decoder_decode_s": (i32, [C.POINTER(LdpcDecoder), vp, vp, u32]),
            "srsran_ldpc_decoder_decode_crc_c": (i32, [C.POINTER(LdpcDecoder), vp, vp, u32, C.POINTER(Crc)]),
            "create_compact_pcm": (i32, [vp, vp, i32, C.c_uint16]),
            "srsran_ofdm_rx_init_cfg": (i32, [C.POINTER(Ofdm), C.POINTER(OfdmCfg)]),
            "srsran_ofdm_tx_init_cfg": (i32, [C.POINTER(Ofdm), C.POINTER(OfdmCfg)]),
            "srsran_ofdm_rx_init": (i32, [C.POINTER(Ofdm), i32, vp, vp, u32]),
            "srsran_ofdm_tx_init": (i32, [C.POINTER(Ofdm), i32, vp, vp, u32]),
            "srsran_ofdm_rx_set_prb": (i32, [C.POINTER(Ofdm), i32, u32]),
            "srsran_ofdm_tx_set_prb": (i32, [C.POINTER(Ofdm), i32, u32]),
            "srsran_ofdm_rx_free": (None, [C.POINTER(Ofdm)]),
            "srsran_ofdm_tx_free": (None, [C.POINTER(Ofdm)]),
            "srsran_ofdm_rx_sf": (None, [C.POINTER(Ofdm)]),
            "srsran_ofdm_rx_sf_ng": (None, [C.POINTER(Ofdm), vp, vp]),
            "srsran_ofdm_tx_sf": (None, [C.POINTER(Ofdm)]),
            "srsran_ofdm_set_freq_shift": (i32, [C.POINTER(Ofdm), C.c_float]),
            "srsran_ofdm_set_normalize": (None, [C.POINTER(Ofdm), C.c_bool]),
            "srsran_dft_plan_c": (i32, [C.POINTER(DftPlan), i32, i32]),
            "srsran_dft_plan_r": (i32, [C.POINTER(DftPlan), i32, i32]),
            "srsran_dft_replan_r": (i32, [C.POINTER(DftPlan), i32]),
            "srsran_dft_run_r": (None, [C.POINTER(DftPlan), vp, vp]),
            "srsran_dft_plan": (i32, [C.POINTER(DftPlan), i32, i32, i32]),
            "srsran_dft_plan_guru_c": (i32, [C.POINTER(DftPlan), i32, i32, vp, vp, i32, i32, i32, i32, i32]),
            "srsran_dft_replan": (i32, [C.POINTER(DftPlan), i32]),
            "srsran_dft_replan_c": (i32, [C.POINTER(DftPlan), i32]),
            "srsran_dft_plan_free": (None, [C.POINTER(DftPlan)]),
            "srsran_dft_plan_set_mirror": (None, [C.POINTER(DftPlan), C.c_bool]),
            "srsran_dft_plan_set_db": (None, [C.POINTER(DftPlan), C.c_bool]),
            "srsran_dft_plan_set_norm": (None, [C.POINTER(DftPlan), C.c_bool]),
            "srsran_dft_plan_set_dc": (None, [C.POINTER(DftPlan), C.c_bool]),
            "srsran_dft_run_c": (None, [C.POINTER(DftPlan), vp, vp]),
            "srsran_dft_run_c_zerocopy": (None, [C.POINTER(DftPlan), vp, vp]),
            "srsran_dft_run_guru_c": (None, [C.POINTER(DftPlan)]),
            "srsran_dft_precoding_init": (i32, [C.POINTER(DftPrecoding), u32, C.c_bool]),
            "srsran_dft_precoding_free": (None, [C.POINTER(DftPrecoding)]),
            "srsran_dft_precoding_valid_prb": (C.c_bool, [u32]),
            "srsran_dft_precoding_get_valid_prb": (u32, [u32]),
            "srsran_dft_precoding": (i32, [C.POINTER(DftPrecoding), vp, vp, u32, u32]),
            "srsran_hip_dft_batch_create": (i32, [C.POINTER(vp), i32, i32, C.c_bool, C.c_bool, C.c_bool]),
            "srsran_hip_dft_batch_free": (None, [vp]),
            "srsran_hip_dft_batch_run": (i32, [vp, vp, vp, u32, vp]),
            "srsran_pss_init_fft": (i32, [C.POINTER(Pss), u32, u32]),
            "srsran_pss_init_fft_offset": (i32, [C.POINTER(Pss), u32, u32, i32]),
            "srsran_pss_init_fft_offset_decim": (i32, [C.POINTER(Pss), u32, u32, i32, i32]),
            "srsran_pss_init": (i32, [C.POINTER(Pss), u32]),
            "srsran_pss_resize": (i32, [C.POINTER(Pss), u32, u32, i32]),
            "srsran_pss_free": (None, [C.POINTER(Pss)]),
            "srsran_pss_reset": (None, [C.POINTER(Pss)]),
            "srsran_pss_generate": (i32, [vp, u32]),
            "srsran_pss_put_slot": (None, [vp, vp, u32, i32]),
            "srsran_pss_get_slot": (None, [vp, vp, u32, i32]),
            "srsran_pss_set_ema_alpha": (None, [C.POINTER(Pss), C.c_float]),
            "srsran_pss_set_N_id_2": (i32, [C.POINTER(Pss), u32]),
            "srsran_pss_find_pss": (i32, [C.POINTER(Pss), vp, C.POINTER(C.c_float)]),
            "srsran_sss_init": (i32, [C.POINTER(Sss), u32]),
            "srsran_sss_resize": (i32, [C.POINTER(Sss), u32]),
            "srsran_sss_free": (None, [C.POINTER(Sss)]),
            "srsran_sss_generate": (None, [vp, vp, u32]),
            "srsran_sss_put_slot": (None, [vp, vp, u32, i32]),
            "srsran_sss_set_N_id_2": (i32, [C.POINTER(Sss), u32]),
            "srsran_sss_set_threshold": (None, [C.POINTER(Sss), C.c_float]),
            "srsran_sss_m0m1_partial": (i32, [C.POINTER(Sss), vp, u32, vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(u32), C.POINTER(C.c_float)]),
            "srsran_sss_m0m1_diff": (i32, [C.POINTER(Sss), vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(u32), C.POINTER(C.c_float)]),
            "srsran_sss_m0m1_diff_coh": (i32, [C.POINTER(Sss), vp, vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(u32), C.POINTER(C.c_float)]),
            "srsran_sss_subframe": (u32, [u32, u32]),
            "srsran_sss_N_id_1": (i32, [C.POINTER(Sss), u32, u32, C.c_float]),
            "srsran_pss_filter_enable": (None, [C.POINTER(Pss), C.c_bool]),
            "srsran_pss_filter": (None, [C.POINTER(Pss), vp, vp]),
            "srsran_pss_chest": (i32, [C.POINTER(Pss), vp, vp]),
            "srsran_pss_cfo_compute": (C.c_float, [C.POINTER(Pss), vp]),
            "srsran_pss_sic": (None, [C.POINTER(Pss), vp]),
            "srsran_cfo_init": (i32, [C.POINTER(Cfo), u32]),
            "srsran_cfo_free": (None, [C.POINTER(Cfo)]),
            "srsran_cfo_resize": (i32, [C.POINTER(Cfo), u32]),
            "srsran_cfo_set_tol": (None, [C.POINTER(Cfo), C.c_float]),
            "srsran_cfo_correct": (None, [C.POINTER(Cfo), vp, vp, C.c_float]),
            "srsran_cfo_correct_offset": (None, [C.POINTER(Cfo), vp, vp, C.c_float, i32, i32]),
            "srsran_cp_synch_init": (i32, [C.POINTER(CpSynch), u32]),
            "srsran_cp_synch_free": (None, [C.POINTER(CpSynch)]),
            "srsran_cp_synch_resize": (i32, [C.POINTER(CpSynch), u32]),
            "srsran_cp_synch": (u32, [C.POINTER(CpSynch), vp, u32, u32, u32]),
            "srsran_sync_init": (i32, [C.POINTER(Sync), u32, u32, u32]),
            "srsran_sync_init_decim": (i32, [C.POINTER(Sync), u32, u32, u32, i32]),
            "srsran_sync_free": (None, [C.POINTER(Sync)]),
            "srsran_sync_resize": (i32, [C.POINTER(Sync), u32, u32, u32]),
            "srsran_sync_reset": (None, [C.POINTER(Sync)]),
            "srsran_sync_find": (i32, [C.POINTER(Sync), vp, u32, C.POINTER(u32)]),
            "srsran_sync_detect_cp": (i32, [C.POINTER(Sync), vp, u32]),
            "srsran_sync_set_threshold": (None, [C.POINTER(Sync), C.c_float]),
            "srsran_sync_get_sf_idx": (u32, [C.POINTER(Sync)]),
            "srsran_sync_get_peak_value": (C.c_float, [C.POINTER(Sync)]),
            "srsran_sync_set_sss_algorithm": (None, [C.POINTER(Sync), i32]),
            "srsran_sync_set_em_alpha": (None, [C.POINTER(Sync), C.c_float]),
            "srsran_sync_set_N_id_2": (i32, [C.POINTER(Sync), u32]),
            "srsran_sync_set_N_id_1": (i32, [C.POINTER(Sync), u32]),
            "srsran_sync_get_cell_id": (i32, [C.POINTER(Sync)]),
            "srsran_sync_set_pss_filt_enable": (None, [C.POINTER(Sync), C.c_bool]),
            "srsran_sync_set_sss_eq_enable": (None, [C.POINTER(Sync), C.c_bool]),
            "srsran_sync_get_cfo": (C.c_float, [C.POINTER(Sync)]),
            "srsran_sync_cfo_reset": (None, [C.POINTER(Sync), C.c_float]),
            "srsran_sync_copy_cfo": (None, [C.POINTER(Sync), C.POINTER(Sync)]),
            "srsran_sync_set_cfo_i_enable": (None, [C.POINTER(Sync), C.c_bool]),
            "srsran_sync_set_cfo_cp_enable": (None, [C.POINTER(Sync), C.c_bool, u32]),
            "srsran_sync_set_cfo_pss_enable": (None, [C.POINTER(Sync), C.c_bool]),
            "srsran_sync_set_cfo_tol": (None, [C.POINTER(Sync), C.c_float]),
            "srsran_sync_set_frame_type": (None, [C.POINTER(Sync), i32]),
            "srsran_sync_set_cfo_ema_alpha": (None, [C.POINTER(Sync), C.c_float]),
            "srsran_sync_get_cp": (i32, [C.POINTER(Sync)]),
            "srsran_sync_set_cp": (None, [C.POINTER(Sync), i32]),
            "srsran_sync_sss_en": (None, [C.POINTER(Sync), C.c_bool]),
            "srsran_sync_get_cur_pss_obj": (C.POINTER(Pss), [C.POINTER(Sync)]),
            "srsran_sync_sss_detected": (C.c_bool, [C.POINTER(Sync)]),
            "srsran_sync_sss_correlation_peak": (C.c_float, [C.POINTER(Sync)]),
            "srsran_sync_sss_available": (C.c_bool, [C.POINTER(Sync)]),
            "srsran_sync_cp_en": (None, [C.POINTER(Sync), C.c_bool]),
            "srsran_rm_turbo_rx_lut": (i32, [vp, vp, u32, u32, u32]),
            "srsran_rm_turbo_rx_lut_": (i32, [vp, vp, u32, u32, u32, C.c_bool]),
            "srsran_rm_turbo_rx_lut_8bit": (i32, [vp, vp, u32, u32, u32]),
            "srsran_rm_turbo_gentables": (None, []),
            "srsran_rm_turbo_free_tables": (None, []),
            "srsran_hip_rm_turbo_table": (i32, [vp, u32, u32, u32]),
            "srsran_hip_rm_turbo_rx_batch": (i32, [vp, u32, u32, vp, u32, u32, u32, u32, u32, vp]),
            "srsran_hip_rm_turbo_rx_batch_8bit": (i32, [vp, u32, u32, vp, u32, u32, u32, u32, u32, vp]),
            "srsran_hip_sch_create": (i32, [C.POINTER(vp)]),
            "srsran_hip_sch_free": (None, [vp]),
            "srsran_hip_sch_decode": (i32, [vp, vp, C.POINTER(HipTb), u32, u32, vp, vp, vp, C.POINTER(HipTbResult), vp]),
            "srsran_hip_sch_decode_8bit": (i32, [vp, vp, C.POINTER(HipTb), u32, u32, vp, vp, vp, C.POINTER(HipTbResult), vp]),
            "srsran_cbsegm": (i32, [C.POINTER(Cbsegm), u32]),
            "srsran_tcod_init": (i32, [C.POINTER(Tcod), u32]),
            "srsran_tcod_free": (None, [C.POINTER(Tcod)]),
            "srsran_tcod_encode": (i32, [C.POINTER(Tcod), vp, vp, u32]),
            "srsran_hip_tcod_encode_batch": (i32, [vp, u32, vp, u32, u32, u32, vp]),
            "srsran_tcod_gentable": (None, []),
            "srsran_tcod_encode_lut": (i32, [C.POINTER(Tcod), C.POINTER(Crc), C.POINTER(Crc), vp, vp, u32, C.c_bool]),
            "srsran_rm_turbo_tx_lut": (i32, [vp, vp, vp, vp, u32, u32, u32, u32]),
            "srsran_hip_sch_enc_create": (i32, [C.POINTER(vp)]),
            "srsran_hip_sch_enc_free": (None, [vp]),
            "srsran_hip_sch_encode": (i32, [vp, vp, C.POINTER(HipTb), u32, vp, vp]),
            "srsran_ldpc_rm_tx_init": (i32, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_rx_init_f": (i32, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_rx_init_s": (i32, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_rx_init_c": (i32, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_tx_free": (None, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_rx_free_f": (None, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_rx_free_s": (None, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_rx_free_c": (None, [C.POINTER(LdpcRm)]),
            "srsran_ldpc_rm_tx": (i32, [C.POINTER(LdpcRm), vp, vp, u32, i32, u32, C.c_uint8, i32, u32]),
            "srsran_ldpc_rm_rx_f": (i32, [C.POINTER(LdpcRm), vp, vp, u32, u32, i32, u32, C.c_uint8, i32, u32]),
            "srsran_ldpc_rm_rx_s": (i32, [C.POINTER(LdpcRm), vp, vp, u32, u32, i32, u32, C.c_uint8, i32, u32]),
            "srsran_ldpc_rm_rx_c": (i32, [C.POINTER(LdpcRm), vp, vp, u32, u32, i32, u32, C.c_uint8, i32, u32]),
            "srsran_ldpc_encoder_init": (i32, [C.POINTER(LdpcEncoder), i32, i32, C.c_uint16]),
            "srsran_ldpc_encoder_free": (None, [C.POINTER(LdpcEncoder)]),
            "srsran_ldpc_encoder_encode": (i32, [C.POINTER(LdpcEncoder), vp, vp, u32]),
            "srsran_ldpc_encoder_encode_rm": (i32, [C.POINTER(LdpcEncoder), vp, vp, u32, u32]),
            "srsran_hip_nr_sch_create": (i32, [C.POINTER(vp)]),
            "srsran_hip_nr_sch_free": (None, [vp]),
            "srsran_hip_ldpc_rm_rx_batch": (i32, [vp, i32, vp, vp, C.POINTER(HipLdpcCb), u32, u32, i32, u32, u32, i32, u32, vp]),
            "srsran_hip_ldpc_rm_rx_batch_new": (i32, [vp, i32, vp, vp, C.POINTER(HipLdpcCb), vp, u32, u32, i32, u32, u32, i32, u32, vp]),
            "srsran_hip_ldpc_rm_tx_batch": (i32, [vp, vp, vp, C.POINTER(HipLdpcCb), u32, i32, u32, u32, i32, u32, vp]),
            "srsran_hip_ldpc_encode_batch": (i32, [vp, vp, vp, C.POINTER(HipLdpcCb), u32, i32, u32, vp]),
            "srsran_predecoding_single": (i32, [vp, vp, vp, vp, i32, C.c_float, C.c_float]),
            "srsran_hip_predecoding_single": (i32, [vp, vp, vp, vp, u32, C.c_float, C.c_float, vp]),
            "srsran_demod_soft_demodulate": (i32, [i32, vp, vp, i32]),
            "srsran_demod_soft_demodulate_s": (i32, [i32, vp, vp, i32]),
            "srsran_demod_soft_demodulate_b": (i32, [i32, vp, vp, i32]),
            "srsran_sequence_apply_f": (None, [vp, vp, u32, u32]),
            "srsran_sequence_apply_s": (None, [vp, vp, u32, u32]),
            "srsran_sequence_apply_c": (None, [vp, vp, u32, u32]),
            "srsran_sequence_pdsch_apply_f": (None, [vp, vp, C.c_uint16, i32, u32, u32, u32]),
            "srsran_sequence_pdsch_apply_s": (None, [vp, vp, C.c_uint16, i32, u32, u32, u32]),
            "srsran_sequence_pdsch_apply_c": (None, [vp, vp, C.c_uint16, i32, u32, u32, u32]),
            "srsran_sequence_pusch_apply_s": (None, [vp, vp, C.c_uint16, u32, u32, u32]),
            "srsran_sequence_pusch_apply_c": (None, [vp, vp, C.c_uint16, u32, u32, u32]),
            "srsran_hip_demod_create": (i32, [C.POINTER(vp)]),
            "srsran_hip_demod_free": (None, [vp]),
            "srsran_hip_demod_run": (i32, [vp, vp, vp, i32, C.POINTER(HipDemodJob), u32, vp]),
            "srsran_hip_sequence_pdsch_seed": (u32, [C.c_uint16, i32, u32, u32]),
            "srsran_hip_sequence_pusch_seed": (u32, [C.c_uint16, u32, u32]),
            "srsran_hip_pusch_decode": (i32, [C.POINTER(HipPuschRx), vp, vp, C.POINTER(SoftbufferRx), vp, C.POINTER(HipGrantRes)]),
            "srsran_hip_pusch_decode_multi": (i32, [u32, C.POINTER(HipPuschRx), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.POINTER(SoftbufferRx)),
                                                    C.POINTER(vp), C.POINTER(HipGrantRes)]),
            "srsran_hip_pdsch_decode": (i32, [C.POINTER(HipPdschRx), vp, vp, C.POINTER(SoftbufferRx), vp, C.POINTER(HipGrantRes)]),
            "srsran_hip_pdsch_encode": (i32, [C.POINTER(HipPdschTx), C.POINTER(SoftbufferTx), vp, vp]),
            "srsran_hip_pdsch_decode_dbg": (i32, [C.POINTER(HipPdschRx), vp, vp, C.POINTER(SoftbufferRx), vp, C.POINTER(HipGrantRes), vp, vp]),
            "srsran_hip_pdsch_encode_dbg": (i32, [C.POINTER(HipPdschTx), C.POINTER(SoftbufferTx), vp, vp, vp]),
            "srsran_hip_pdsch_encode_multi": (i32, [u32, C.POINTER(HipPdschTx), C.POINTER(C.POINTER(SoftbufferTx)), C.POINTER(vp), C.POINTER(vp)]),
            "srsran_hip_ulsch_encode": (i32, [C.POINTER(HipGrantTb), u32, C.POINTER(SoftbufferTx), vp, vp]),
            "srsran_hip_modulate_bytes": (i32, [u32, vp, vp, u32, u32, u32, C.c_float]),
            "srsran_hip_cellsearch_create": (i32, [C.POINTER(vp), u32, u32, i32, i32, u32]),
            "srsran_hip_cellsearch_free": (None, [vp]),
            "srsran_hip_cellsearch_run": (i32, [vp, vp, u32, i32, vp, vp]),
            "srsran_hip_cellsearch_corr": (vp, [vp, u32, u32]),
            "srsran_symbol_sz": (i32, [u32]),
            "srsran_symbol_sz_power2": (i32, [u32]),
            "srsran_use_standard_symbol_size": (None, [C.c_bool]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().srsran_hip_last_error().decode()


def check(rc, what):
    if rc != SRSRAN_SUCCESS:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, last_error()))
