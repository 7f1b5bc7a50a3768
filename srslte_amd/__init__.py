"""srslte_amd -- MI355X-native PHY DSP engine behind the srsRAN `srsran_*_t` C ABI.

The product is `lib/libsrsran_phy_hip.so` (hand-written HIP kernels for gfx950 + C-ABI host layer,
sources in `csrc/`).  This package is only a thin ctypes mirror of that ABI for tests, bench and
Python callers: no computation happens in Python and there is no CPU fallback.
"""
import ctypes as C
import os

# Worker threads overlap only when their streams sit on different hardware queues (csrc/common.cpp): the runtime reads GPU_MAX_HW_QUEUES when it
# initialises, which in a Python process is often before the library is loaded (torch) -- so the package asks here, at import, unless a value is set.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

from . import capi  # noqa: E402
from .capi import lib  # noqa: F401,E402

__all__ = ["capi", "lib", "DeviceBuffer", "TdecBatch", "LdpcBatch", "OfdmBatch"]


def _ptr(x):
    """device pointer from an int, a DeviceBuffer, or anything with .data_ptr() (torch tensor)"""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class DeviceBuffer:
    """hipMalloc'ed buffer owned through the C ABI (srsran_hip_malloc)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().srsran_hip_malloc(self.nbytes)
        if not self.ptr:
            raise RuntimeError("srsran_hip_malloc(%d) failed: %s" % (nbytes, capi.last_error()))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        capi.check(lib().srsran_hip_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, None), "memcpy_h2d")
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        capi.check(lib().srsran_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib().srsran_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TdecBatch:
    """srsran_tdec_run_all over a batch of code blocks (srsran_hip_tdec_batch_*, phy_batch.h)."""

    def __init__(self, long_cb, max_nof_cb, impl=capi.TDEC_AUTO, llr8=False):
        """llr8: resolve AUTO as the 8-bit API does (srsran_tdec_run_all_8bit); decode() then takes int8 LLRs"""
        self.K, self.max_cb, self.llr8 = int(long_cb), int(max_nof_cb), bool(llr8)
        self._h = C.c_void_p()
        create = lib().srsran_hip_tdec_batch_create_8bit if llr8 else lib().srsran_hip_tdec_batch_create
        capi.check(create(C.byref(self._h), self.K, self.max_cb, impl), "tdec_batch_create")

    def run(self, d_input, in_stride, d_output, out_stride, n_cb, nof_iterations, sb_layout=0, stream=None):
        capi.check(lib().srsran_hip_tdec_batch_run(self._h, _ptr(d_input), in_stride, _ptr(d_output), out_stride, n_cb,
                                                   nof_iterations, sb_layout, stream), "tdec_batch_run")

    def decode(self, llr, nof_iterations, sb_layout=0, want_llr=False, n_begin=0):
        """host convenience: llr int16 (or int8) [n_cb, L] -> packed bytes [n_cb, K/8] (and decision LLRs)"""
        in8 = np.asarray(llr).dtype == np.int8
        llr = np.ascontiguousarray(llr, dtype=np.int8 if in8 else np.int16)
        n_cb, L = llr.shape
        d_in = DeviceBuffer.from_numpy(llr)
        d_out = DeviceBuffer(n_cb * (self.K // 8))
        if want_llr:
            dbg = lib().srsran_hip_tdec_batch_run_dbg_8bit if in8 else lib().srsran_hip_tdec_batch_run_dbg
            capi.check(dbg(self._h, d_in.ptr, L, d_out.ptr, self.K // 8, n_cb, n_begin, max(1, nof_iterations), sb_layout, None),
                       "tdec_batch_run_dbg")
        elif in8:
            capi.check(lib().srsran_hip_tdec_batch_run_8bit(self._h, d_in.ptr, L, d_out.ptr, self.K // 8, n_cb, nof_iterations,
                                                            sb_layout, None), "tdec_batch_run_8bit")
        else:
            self.run(d_in, L, d_out, self.K // 8, n_cb, nof_iterations, sb_layout)
        capi.check(lib().srsran_hip_stream_sync(None), "sync")
        out = d_out.to_numpy(np.uint8, (n_cb, self.K // 8))
        if want_llr:
            d_l = DeviceBuffer(n_cb * self.K * 2)
            capi.check(lib().srsran_hip_tdec_batch_last_llr(self._h, d_l.ptr, n_cb, None), "last_llr")
            capi.check(lib().srsran_hip_stream_sync(None), "sync")
            return out, d_l.to_numpy(np.int16, (n_cb, self.K))
        return out

    def free(self):
        if self._h:
            lib().srsran_hip_tdec_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class LdpcBatch:
    """srsran_ldpc_decoder_decode_c over a batch of code words (srsran_hip_ldpc_batch_*)."""

    def __init__(self, bg, ls, scaling_fctr=0.8, max_nof_iter=0, max_nof_cw=1, dec_type=capi.LDPC_C):
        """dec_type: capi.LDPC_C (int8 LLRs), LDPC_S (int16) or LDPC_F (float32)"""
        self.bg, self.Z = int(bg), int(ls)
        self.bgN, self.bgK = (68, 22) if bg == capi.BG1 else (52, 10)
        self.max_iter = int(max_nof_iter) if max_nof_iter else 10
        if dec_type in (capi.LDPC_C_FLOOD, capi.LDPC_C_AVX2_FLOOD, capi.LDPC_C_AVX512_FLOOD):
            self.max_iter *= 2  # the flooded schedule runs twice the iterations (ldpc_decoder.c:136)
        self.dtype = {capi.LDPC_F: np.float32, capi.LDPC_S: np.int16}.get(dec_type, np.int8)
        self._h = C.c_void_p()
        capi.check(lib().srsran_hip_ldpc_batch_create_typed(C.byref(self._h), self.bg, self.Z, scaling_fctr, max_nof_iter,
                                                            max_nof_cw, dec_type), "ldpc_batch_create")

    @property
    def n_llr(self):
        return (self.bgN - 2) * self.Z

    @property
    def liftK(self):
        return self.bgK * self.Z

    def run(self, d_llrs, llr_stride, d_msg, msg_stride, n_cw, cdwd_rm_length, d_iter_msgs=None, stream=None):
        capi.check(lib().srsran_hip_ldpc_batch_run_typed(self._h, _ptr(d_llrs), llr_stride, _ptr(d_msg), msg_stride, n_cw,
                                                         cdwd_rm_length, _ptr(d_iter_msgs), stream), "ldpc_batch_run")

    def decode(self, llrs, cdwd_rm_length=None, want_iter_msgs=False, want_soft=False):
        llrs = np.ascontiguousarray(llrs, dtype=self.dtype)
        n_cw, L = llrs.shape
        assert L >= self.n_llr
        d_in = DeviceBuffer.from_numpy(llrs)
        d_out = DeviceBuffer(n_cw * self.liftK)
        mb = (self.liftK + 7) // 8
        d_it = DeviceBuffer(n_cw * self.max_iter * mb) if want_iter_msgs else None
        rm = self.n_llr if cdwd_rm_length is None else cdwd_rm_length
        if want_soft:
            d_soft = DeviceBuffer(n_cw * self.bgN * self.Z * np.dtype(self.dtype).itemsize)
            capi.check(lib().srsran_hip_ldpc_batch_run_dbg(self._h, d_in.ptr, L, d_out.ptr, self.liftK, n_cw, rm, d_soft.ptr, None),
                       "ldpc_batch_run_dbg")
            capi.check(lib().srsran_hip_stream_sync(None), "sync")
            return d_out.to_numpy(np.uint8, (n_cw, self.liftK)), d_soft.to_numpy(self.dtype, (n_cw, self.bgN * self.Z))
        self.run(d_in, L, d_out, self.liftK, n_cw, rm, d_it)
        capi.check(lib().srsran_hip_stream_sync(None), "sync")
        out = d_out.to_numpy(np.uint8, (n_cw, self.liftK))
        if want_iter_msgs:
            return out, d_it.to_numpy(np.uint8, (n_cw, self.max_iter, mb))
        return out

    def free(self):
        if self._h:
            lib().srsran_hip_ldpc_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class OfdmBatch:
    """srsran_ofdm_rx_sf / srsran_ofdm_tx_sf over a batch of subframes (srsran_hip_ofdm_batch_*)."""

    def __init__(self, nof_prb, tx=False, symbol_sz=0, cp=capi.CP_NORM, normalize=False, freq_shift_f=0.0,
                 rx_window_offset=0.0, keep_dc=False, mbsfn_region=0):
        """mbsfn_region: 0 = normal subframes; 1, 2 = MBSFN subframes with that non-MBSFN region (needs CP_EXT)"""
        cfg = capi.OfdmCfg()
        cfg.nof_prb, cfg.cp, cfg.sf_type = nof_prb, cp, (capi.SF_MBSFN if mbsfn_region else capi.SF_NORM)
        cfg.normalize, cfg.freq_shift_f, cfg.rx_window_offset = normalize, freq_shift_f, rx_window_offset
        cfg.symbol_sz, cfg.keep_dc = symbol_sz, keep_dc
        self.tx = tx
        self._h = C.c_void_p()
        capi.check(lib().srsran_hip_ofdm_batch_create(C.byref(self._h), C.byref(cfg),
                                                      capi.DFT_BACKWARD if tx else capi.DFT_FORWARD), "ofdm_batch_create")
        if mbsfn_region:
            capi.check(lib().srsran_hip_ofdm_batch_set_non_mbsfn_region(self._h, mbsfn_region), "set_non_mbsfn_region")
        self.sf_sz = lib().srsran_hip_ofdm_batch_sf_sz(self._h)
        self.sf_re = lib().srsran_hip_ofdm_batch_sf_re(self._h)

    def run(self, d_in, d_out, n_sf, stream=None):
        f = lib().srsran_hip_ofdm_batch_tx if self.tx else lib().srsran_hip_ofdm_batch_rx
        capi.check(f(self._h, _ptr(d_in), _ptr(d_out), n_sf, stream), "ofdm_batch_run")

    def process(self, x):
        """host convenience: complex64 [n_sf, sf_sz] -> [n_sf, sf_re] (rx) or the reverse (tx)"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n_sf = x.shape[0]
        n_in, n_out = (self.sf_re, self.sf_sz) if self.tx else (self.sf_sz, self.sf_re)
        assert x.shape[1] == n_in
        d_in = DeviceBuffer.from_numpy(x)
        d_out = DeviceBuffer(n_sf * n_out * 8)
        capi.check(lib().srsran_hip_memset(d_out.ptr, 0, d_out.nbytes, None), "memset")
        self.run(d_in, d_out, n_sf)
        capi.check(lib().srsran_hip_stream_sync(None), "sync")
        return d_out.to_numpy(np.complex64, (n_sf, n_out))

    def free(self):
        if self._h:
            lib().srsran_hip_ofdm_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
