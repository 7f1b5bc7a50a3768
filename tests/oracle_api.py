"""ctypes access to oracle/liboracle.so (the CPU restatement of the reference) for the tests.

TEST INFRASTRUCTURE ONLY: nothing in srslte_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libsrsran_ref.so")

ORC_TDEC_AUTO, ORC_TDEC_GENERIC, ORC_TDEC_SSE_WINDOW, ORC_TDEC_AVX_WINDOW, ORC_TDEC_SSE8_WINDOW, ORC_TDEC_AVX8_WINDOW = 0, 1, 3, 5, 6, 7


class LdpcGraph(C.Structure):
    _fields_ = [("bg", C.c_int), ("ls", C.c_uint16), ("bgN", C.c_int), ("bgM", C.c_int), ("bgK", C.c_int),
                ("nof_edges", C.c_int), ("row_start", C.c_uint16 * 47), ("col", C.c_uint8 * 320),
                ("shift", C.c_uint16 * 320)]


class OfdmCfg(C.Structure):
    _fields_ = [("nof_prb", C.c_uint32), ("symbol_sz", C.c_uint32), ("cp_ext", C.c_int), ("normalize", C.c_int),
                ("freq_shift_f", C.c_float), ("rx_window_offset", C.c_float), ("keep_dc", C.c_int), ("mbsfn_region", C.c_int)]


_orc = None


def build_oracle():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(ORACLE_LIB) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return ORACLE_LIB


def orc():
    global _orc
    if _orc is None:
        L = C.CDLL(build_oracle())
        vp = C.c_void_p
        L.orc_tdec_run_all.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, vp]
        L.orc_tdec_run_all_8bit.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp]
        L.orc_tcod_encode.argtypes = [vp, vp, C.c_uint32]
        L.orc_qpp_gen.argtypes = [C.c_uint32, C.c_uint32, vp, vp]
        L.orc_ldpc_graph.argtypes = [C.POINTER(LdpcGraph), C.c_int, C.c_uint16]
        L.orc_ldpc_decode_c.argtypes = [C.POINTER(LdpcGraph), C.c_float, C.c_int, vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp]
        L.orc_ldpc_encode.argtypes = [C.POINTER(LdpcGraph), vp, vp]
        L.orc_crc_bits.argtypes = [C.c_uint32, C.c_int, vp, C.c_int]
        L.orc_crc_bits.restype = C.c_uint32
        L.orc_ofdm_rx_sf.argtypes = [C.POINTER(OfdmCfg), vp, vp]
        L.orc_ofdm_tx_sf.argtypes = [C.POINTER(OfdmCfg), vp, vp]
        L.orc_dft_c.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _orc = L
    return _orc


def have_ref():
    return os.path.exists(REF_LIB)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ turbo helpers
def tc_sizes():
    return [orc().orc_tc_cb_size(i) for i in range(188)]


def turbo_encode(bits):
    K = bits.size
    out = np.zeros(3 * K + 12, np.uint8)
    assert orc().orc_tcod_encode(P(np.ascontiguousarray(bits, np.uint8)), P(out), K) == 0
    return out


def turbo_llrs(K, n_cb, esn0_db, seed, scale=100.0):
    """random messages -> turbo code -> BPSK -> AWGN -> int16 LLR = round(scale*y)  (turbodecoder_test.c:254)"""
    rng = np.random.default_rng(seed)
    msgs = rng.integers(0, 2, (n_cb, K)).astype(np.uint8)
    llr = np.zeros((n_cb, 3 * K + 12), np.int16)
    sigma = 10 ** (-esn0_db / 20)
    for i in range(n_cb):
        enc = turbo_encode(msgs[i])
        y = (2.0 * enc - 1.0) + sigma * rng.standard_normal(enc.size)
        llr[i] = np.clip(np.round(scale * y), -32768, 32767).astype(np.int16)
    return msgs, llr


def turbo_decode(llr, nof_iterations, K, impl=ORC_TDEC_AUTO, sb_layout=0, want_llr=False):
    llr = np.ascontiguousarray(llr, np.int16)
    n_cb = llr.shape[0]
    out = np.zeros((n_cb, K // 8), np.uint8)
    dl = np.zeros((n_cb, K), np.int16)
    for i in range(n_cb):
        rc = orc().orc_tdec_run_all(P(llr[i]), P(out[i]), nof_iterations, K, impl, sb_layout, None, P(dl[i]))
        assert rc == 0, rc
    return (out, dl) if want_llr else out


def turbo_decode_8bit(llr, nof_iterations, K, impl=ORC_TDEC_AUTO, sb_layout=0, want_llr=False):
    """srsran_tdec_run_all_8bit on int8 LLRs"""
    llr = np.ascontiguousarray(llr, np.int8)
    n_cb = llr.shape[0]
    out = np.zeros((n_cb, K // 8), np.uint8)
    dl = np.zeros((n_cb, K), np.int16)
    for i in range(n_cb):
        rc = orc().orc_tdec_run_all_8bit(P(llr[i]), P(out[i]), nof_iterations, K, impl, sb_layout, P(dl[i]))
        assert rc == 0, rc
    return (out, dl) if want_llr else out


def turbo_llrs_8bit(K, n_cb, esn0_db, seed, scale=12.0):
    """int8 LLRs for the 8-bit API (scaled to about +-30, clipped to +-127)"""
    msgs, llr = turbo_llrs(K, n_cb, esn0_db, seed, scale=scale)
    return msgs, np.clip(llr, -127, 127).astype(np.int8)


CRC24A, CRC24B = 0x1864CFB, 0x1800063


def crc_attach(bits, poly):
    c = orc().orc_crc_bits(poly, 24, P(np.ascontiguousarray(bits, np.uint8)), bits.size)
    return np.concatenate([bits, np.array([(c >> (23 - i)) & 1 for i in range(24)], np.uint8)])


def cbsegm(tbs):
    v = [C.c_uint32() for _ in range(6)]
    orc().orc_cbsegm.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint32)] * 6
    assert orc().orc_cbsegm(tbs, *[C.byref(x) for x in v]) == 0
    return dict(zip(("C", "K1", "K2", "C1", "C2", "F"), [x.value for x in v]))


def rm_table(K, rv, nsb=0):
    orc().orc_rm_turbo_deinter.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    t = np.zeros(3 * K + 12, np.uint16)
    assert orc().orc_rm_turbo_deinter(P(t), K, rv, nsb) == 0
    return t


def tb_coded_bits(tbs, Qm, nof_e_bits, rv, rng, payload=None, tx_order=False):
    """transmit side of one transport block (36.212 5.1.1-5.1.5 as sch.c encode_tb does it: CRC24A, segmentation,
    CRC24B per block, turbo code, rate matching of redundancy version rv, concatenation).
    Returns (coded bits e, uint8 [<= nof_e_bits]; payload bytes incl. the CRC24A)"""
    s = cbsegm(tbs)
    assert s["F"] == 0
    payload = rng.integers(0, 2, tbs).astype(np.uint8) if payload is None else np.asarray(payload, np.uint8)
    b = crc_attach(payload, CRC24A)
    e, pos = [], 0
    Gp = nof_e_bits // Qm
    gamma, n_e = Gp % s["C"], Qm * (Gp // s["C"])
    for i in range(s["C"]):
        # sch.c:392 (receive) takes the C1 blocks of K1 bits first, sch.c:284 (transmit) the C2 blocks of K2 bits; standard
        # transport block sizes have C2 = 0
        K = (s["K2"] if i < s["C2"] else s["K1"]) if tx_order else (s["K1"] if i < s["C1"] else s["K2"])
        rlen = K if s["C"] == 1 else K - 24
        cb = b[pos:pos + rlen]
        pos += rlen
        if s["C"] > 1:
            cb = crc_attach(cb, CRC24B)
        d = turbo_encode(cb)
        E = n_e if i <= s["C"] - gamma - 1 else n_e + (Qm if gamma else 0)  # transmit-side split (sch.c:296-300)
        t = rm_table(K, rv)
        e.append(d[t[np.arange(E) % t.size]])
    return np.concatenate(e).astype(np.uint8), np.packbits(b)


def make_tb(tbs, Qm, nof_e_bits, rv, esn0_db, rng, scale=40.0, payload=None):
    """tb_coded_bits + BPSK over AWGN.  Returns (int16 soft bits e, payload bytes incl. the CRC24A); payload: tbs bits to send again"""
    tx, payload = tb_coded_bits(tbs, Qm, nof_e_bits, rv, rng, payload=payload)
    tx = tx.astype(np.float64)
    sigma = 10 ** (-esn0_db / 20)
    y = (2.0 * tx - 1.0) + sigma * rng.standard_normal(tx.size)
    pad = np.zeros(nof_e_bits - tx.size)
    return np.clip(np.round(scale * np.concatenate([y, pad])), -32768, 32767).astype(np.int16), payload


_AMP = {1: {(): 1}, 2: {(0,): 1, (1,): 3}, 3: {(0, 1): 1, (0, 0): 3, (1, 0): 5, (1, 1): 7},
        4: {(0, 1, 1): 1, (0, 1, 0): 3, (0, 0, 0): 5, (0, 0, 1): 7, (1, 0, 1): 9, (1, 0, 0): 11, (1, 1, 0): 13, (1, 1, 1): 15}}


def modulate(bits, mod):
    """36.211 7.1 constellations, written from the decision rules of demod_soft.c (bit 0/1 of a symbol = sign of I/Q,
    the following pairs select the amplitude); returns complex128 [len(bits) / Qm], unit average power"""
    qm = QM[mod]
    b = np.asarray(bits, np.uint8).reshape(-1, qm)
    if mod == 0:
        return (1 - 2.0 * b[:, 0]) * (1 + 1j) / np.sqrt(2)
    half = qm // 2
    tab = np.zeros(1 << (half - 1))
    for k, v in _AMP[half].items():
        tab[int("".join(map(str, k)) or "0", 2)] = v
    axes = []
    for c in (0, 1):
        idx = np.zeros(b.shape[0], np.int64)
        for j in range(1, half):
            idx = (idx << 1) | b[:, 2 * j + c]
        axes.append((1 - 2.0 * b[:, c]) * tab[idx])
    m = 1 << half
    return (axes[0] + 1j * axes[1]) / np.sqrt(2 * (m * m - 1) / 3)


def sch_decode_tb(tbs, Qm, rv, e_bits, softbuf, cb_crc, max_iterations, cb_data=None):
    """oracle decode_tb: returns (ret, data bytes, avg_iterations); softbuf [C, 18600] int16 and cb_crc [C] uint8 are updated.
    cb_data [C, 768] uint8: decoded code blocks kept between HARQ rounds (softbuffer->data)"""
    f = orc().orc_sch_decode_tb_8bit if e_bits.dtype == np.int8 else orc().orc_sch_decode_tb  # int8: q->llr_is_8bit
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                  C.c_void_p, C.c_void_p]
    data = np.zeros(tbs // 8 + 6, np.uint8)
    cb_data = np.zeros((cb_crc.size, 768), np.uint8) if cb_data is None else cb_data
    avg = C.c_float()
    ret = f(tbs, Qm, rv, e_bits.size, P(e_bits), P(softbuf), P(cb_crc), P(cb_data), max_iterations, P(data), C.byref(avg))
    return ret, data, avg.value


def natural_to_sb_layout(llr_nat, K, nb):
    """what srsran_rm_turbo_rx_lut hands to the window decoders (rm_turbo.c:260-273,
    turbodecoder_iter.h:88-102): syst @0, parity0 @K+32, parity1 @2(K+32) in [step][sub-block] order,
    12 tail LLRs @3(K+32)."""
    out = np.zeros(3 * (K + 32) + 12, llr_nat.dtype)
    sb = K // nb
    n = np.arange(K)
    idx = (n % sb) * nb + n // sb
    out[idx] = llr_nat[3 * n]
    out[K + 32 + idx] = llr_nat[3 * n + 1]
    out[2 * (K + 32) + idx] = llr_nat[3 * n + 2]
    out[3 * (K + 32):] = llr_nat[3 * K:]
    return out


# ------------------------------------------------------------------ LDPC helpers
def ldpc_graph(bg, ls):
    g = LdpcGraph()
    assert orc().orc_ldpc_graph(C.byref(g), bg, ls) == 0
    return g


def ldpc_llrs(bg, ls, n_cw, esn0_db, seed, clip=63):
    """random messages -> LDPC code -> BPSK -> AWGN -> int8 LLR (positive <=> bit 0)"""
    g = ldpc_graph(bg, ls)
    rng = np.random.default_rng(seed)
    K, N = g.bgK * ls, g.bgN * ls
    msgs = rng.integers(0, 2, (n_cw, K)).astype(np.uint8)
    llrs = np.zeros((n_cw, N - 2 * ls), np.int8)
    sigma = 10 ** (-esn0_db / 20)
    for i in range(n_cw):
        cw = np.zeros(N - 2 * ls, np.uint8)
        assert orc().orc_ldpc_encode(C.byref(g), P(msgs[i]), P(cw)) == 0
        y = (1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.size)
        llrs[i] = np.clip(np.round(y * 2.0 / sigma ** 2 * 4), -clip, clip).astype(np.int8)
    return msgs, llrs


def ldpc_decode(bg, ls, llrs, scaling_fctr, max_iter, cdwd_rm_length=None, crc=None):
    g = ldpc_graph(bg, ls)
    K, N = g.bgK * ls, g.bgN * ls
    llrs = np.ascontiguousarray(llrs, np.int8)
    n_cw = llrs.shape[0]
    out = np.zeros((n_cw, K), np.uint8)
    rets = []
    for i in range(n_cw):
        poly, order = crc if crc else (0, 0)
        rets.append(orc().orc_ldpc_decode_c(C.byref(g), scaling_fctr, max_iter, P(llrs[i]), P(out[i]),
                                            N - 2 * ls if cdwd_rm_length is None else cdwd_rm_length, poly, order, None))
    return out, rets


def ldpc_decode_fs(bg, ls, llrs, scaling_fctr, max_iter, cdwd_rm_length=None, want_soft=False):
    """srsran_ldpc_decoder_decode_f (float32 llrs) / _decode_s (int16 llrs)"""
    g = ldpc_graph(bg, ls)
    K, N = g.bgK * ls, g.bgN * ls
    llrs = np.ascontiguousarray(llrs)
    assert llrs.dtype in (np.float32, np.int16)
    fn = orc().orc_ldpc_decode_f if llrs.dtype == np.float32 else orc().orc_ldpc_decode_s
    fn.argtypes = [C.POINTER(LdpcGraph), C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    n_cw = llrs.shape[0]
    out = np.zeros((n_cw, K), np.uint8)
    soft = np.zeros((n_cw, N), llrs.dtype)
    for i in range(n_cw):
        rc = fn(C.byref(g), scaling_fctr, max_iter, P(llrs[i]), P(out[i]), N - 2 * ls if cdwd_rm_length is None else cdwd_rm_length,
                P(soft[i]))
        assert rc == (max_iter if max_iter else 10)
    return (out, soft) if want_soft else out


# ------------------------------------------------------------------ OFDM helpers
def ofdm_cfg(nof_prb, symbol_sz=0, cp_ext=0, normalize=0, freq_shift_f=0.0, rx_window_offset=0.0, keep_dc=0, mbsfn_region=0):
    return OfdmCfg(nof_prb, symbol_sz, cp_ext, normalize, freq_shift_f, rx_window_offset, keep_dc, mbsfn_region)


def ofdm_geometry(cfg):
    N = cfg.symbol_sz if cfg.symbol_sz else orc().orc_symbol_sz(cfg.nof_prb)
    nsym = 12 if cfg.cp_ext else 14
    return N, nsym, 15 * N, nsym * 12 * cfg.nof_prb


def ofdm_rx(cfg, x):
    N, nsym, sf_sz, sf_re = ofdm_geometry(cfg)
    x = np.ascontiguousarray(x, np.complex64)
    out = np.zeros((x.shape[0], sf_re), np.complex64)
    for i in range(x.shape[0]):
        assert orc().orc_ofdm_rx_sf(C.byref(cfg), P(x[i]), P(out[i])) == 0
    return out


def ofdm_tx(cfg, x):
    N, nsym, sf_sz, sf_re = ofdm_geometry(cfg)
    x = np.ascontiguousarray(x, np.complex64)
    out = np.zeros((x.shape[0], sf_sz), np.complex64)
    for i in range(x.shape[0]):
        assert orc().orc_ofdm_tx_sf(C.byref(cfg), P(x[i]), P(out[i])) == 0
    return out


# ------------------------------------------------------------------ sync helpers
def pss_zc(N_id_2):
    zc = np.zeros(62, np.complex64)
    assert orc().orc_pss_generate(P(zc), N_id_2) == 0
    return zc


def sss_seq(cell_id):
    s0, s5 = np.zeros(62, np.float32), np.zeros(62, np.float32)
    assert orc().orc_sss_generate(P(s0), P(s5), cell_id) == 0
    return s0, s5


def sync_subframe(cell_id, nof_prb, symbol_sz, sf5=False):
    """one subframe carrying PSS (last symbol of slot 0) and SSS (the one before), as sync_test.c:130-150 builds it"""
    cfg = ofdm_cfg(nof_prb, symbol_sz, 0, 1)
    grid = np.zeros((14, 12 * nof_prb), np.complex64)
    k = 12 * nof_prb // 2 - 31
    grid[6, k:k + 62] = pss_zc(cell_id % 3)
    grid[5, k:k + 62] = sss_seq(cell_id)[1 if sf5 else 0]
    return ofdm_tx(cfg, grid.reshape(1, -1))[0]


def pss_find(frame, fft_size, N_id_2, want_corr=False, frame_size=None):
    """frame_size < fft_size: the sliding dot-product mode; `frame` must then hold frame_size + fft_size - 1 samples"""
    orc().orc_pss_find.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    frame = np.ascontiguousarray(frame, np.complex64)
    n = frame.size if frame_size is None else frame_size
    pv, psr = C.c_float(), C.c_float()
    corr = np.zeros(n + fft_size - 2, np.float32) if want_corr else None
    pk = orc().orc_pss_find(P(frame), n, fft_size, N_id_2, P(corr) if want_corr else None, C.byref(pv), C.byref(psr))
    return (pk, pv.value, psr.value, corr) if want_corr else (pk, pv.value, psr.value)


def cfo_correct(x, freq):
    orc().orc_cfo_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_uint32]
    x = np.ascontiguousarray(x, np.complex64)
    out = np.zeros_like(x)
    orc().orc_cfo_correct(P(x), P(out), freq, x.size)
    return out


def cp_synch(x, N, max_offset, nof_symbols, cp_len):
    orc().orc_cp_synch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    orc().orc_cp_synch.restype = C.c_uint32
    x = np.ascontiguousarray(x, np.complex64)
    corr = np.zeros(min(max_offset, N), np.complex64)
    idx = orc().orc_cp_synch(P(x), N, max_offset, nof_symbols, cp_len, P(corr))
    return idx, corr


def pss_filter(x, N, N_id_2, want_ce=False):
    orc().orc_pss_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    x = np.ascontiguousarray(x, np.complex64)
    out, ce = np.zeros(N, np.complex64), np.zeros(62, np.complex64)
    assert orc().orc_pss_filter(P(x), P(out), N, N_id_2, P(ce) if want_ce else None) == 0
    return (out, ce) if want_ce else out


def pss_cfo_compute(x, N, N_id_2):
    orc().orc_pss_cfo_compute.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    orc().orc_pss_cfo_compute.restype = C.c_float
    return orc().orc_pss_cfo_compute(P(np.ascontiguousarray(x, np.complex64)), N, N_id_2)


def detect_cp(x, peak_pos, N):
    orc().orc_detect_cp.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    m = np.zeros(2, np.float32)
    cp = orc().orc_detect_cp(P(np.ascontiguousarray(x, np.complex64)), peak_pos, N, P(m))
    return cp, m


def cell_signal(cell_id, nof_prb, symbol_sz, cp_ext=False, tdd=False, sf5=False, n_sf=2):
    """n_sf subframes starting at subframe 0 (or 5) of a cell carrying only PSS and SSS (36.211 6.11): FDD: PSS in
    the last symbol of slot 0, SSS in the one before; TDD: SSS in the last symbol of slot 1, PSS in the third symbol
    of the next subframe.  Returns (samples, index of the first sample after the PSS symbol)."""
    cfg = ofdm_cfg(nof_prb, symbol_sz, 1 if cp_ext else 0, 1)
    N, nsym, sf_sz, sf_re = ofdm_geometry(cfg)
    ns = nsym // 2
    grids = np.zeros((n_sf, nsym, 12 * nof_prb), np.complex64)
    k = 12 * nof_prb // 2 - 31
    if tdd:
        grids[0, nsym - 1, k:k + 62] = sss_seq(cell_id)[1 if sf5 else 0]
        grids[1, 2, k:k + 62] = pss_zc(cell_id % 3)
        pss_sf, pss_sym = 1, 2
    else:
        grids[0, ns - 1, k:k + 62] = pss_zc(cell_id % 3)
        grids[0, ns - 2, k:k + 62] = sss_seq(cell_id)[1 if sf5 else 0]
        pss_sf, pss_sym = 0, ns - 1
    x = ofdm_tx(cfg, grids.reshape(n_sf, -1)).reshape(-1)
    cp0 = orc().orc_cp_len(N, 512 if cp_ext else 160)
    cp1 = orc().orc_cp_len(N, 512 if cp_ext else 144)
    end = pss_sf * sf_sz + cp0 + N + pss_sym * (cp1 + N)
    return x, end


def sss_detect(symbol, fft_size, N_id_2, M):
    orc().orc_sss_m0m1.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    symbol = np.ascontiguousarray(symbol, np.complex64)
    m0, m1, v0, v1, nid, sf = C.c_uint32(), C.c_uint32(), C.c_float(), C.c_float(), C.c_int(), C.c_int()
    assert orc().orc_sss_m0m1(P(symbol), fft_size, N_id_2, M, C.byref(m0), C.byref(v0), C.byref(m1), C.byref(v1), C.byref(nid),
                              C.byref(sf)) == 0
    return m0.value, m1.value, v0.value, v1.value, nid.value, sf.value


def pss_find_fft(frame, fft_size, N_id_2):
    """srsran_pss_find_pss in the reference's algorithmic shape: ONE FFT convolution of length frame + fft per hypothesis
    (pss.c:446-534 -> srsran_conv_fft_cc_run, convolution.c:113-120), |.|^2, first maximum over conv_output_len - 1 entries, PSR --
    on scipy's pocketfft in complex64, one thread.  A CPU-baseline port (the reference's own FFT backend, FFTW, is absent);
    pinned to the direct-sum oracle by test_fft_ports_match_the_oracle.  Returns (peak_pos, peak_value, psr)."""
    import scipy.fft as F

    orc().orc_peak_sidelobe.restype = C.c_float
    orc().orc_peak_sidelobe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    x = np.ascontiguousarray(frame, np.complex64)
    n = x.size
    assert n >= fft_size
    h = np.zeros(fft_size, np.complex64)
    assert orc().orc_pss_time_replica(P(h), N_id_2, fft_size) == 0
    L = n + fft_size
    X = F.fft(x, L, workers=1)
    H = F.fft(h, L, workers=1)
    y = F.ifft(X * H, workers=1)
    avg = np.zeros(L + 2, np.float32)
    avg[:L - 2] = (y.real * y.real + y.imag * y.imag)[:L - 2]
    pk = int(np.argmax(avg[:L - 2]))
    psr = orc().orc_peak_sidelobe(P(avg), pk, L - 1)
    return pk, float(avg[pk]), float(psr)


def ofdm_rx_fft(cfg, x):
    """srsran_ofdm_rx_sf (ofdm.c:395-475) for the plain configuration (normal CP, no frequency shift / window offset / MBSFN) on
    scipy's pocketfft, complex64, one thread: a CPU-baseline port pinned to orc_ofdm_rx by test_fft_ports_match_the_oracle"""
    import scipy.fft as F

    assert cfg.cp_ext == 0 and cfg.freq_shift_f == 0.0 and cfg.rx_window_offset == 0.0 and cfg.mbsfn_region == 0
    N, nsym, sf_sz, sf_re = ofdm_geometry(cfg)
    x = np.ascontiguousarray(x, np.complex64)
    n_sf = x.shape[0]
    nre = sf_re // nsym
    cp0, cp1 = orc().orc_cp_len(N, 160), orc().orc_cp_len(N, 144)
    out = np.empty((n_sf, nsym, nre), np.complex64)
    slot = sf_sz // 2
    starts = [s * slot + cp0 + i * (N + cp1) for s in range(2) for i in range(nsym // 2)]
    sym = np.stack([x[:, st:st + N] for st in starts], axis=1)  # [n_sf, nsym, N]
    Y = F.fft(sym, axis=2, workers=1)
    if cfg.normalize:
        Y = Y * np.float32(1.0 / np.sqrt(N))
    half = nre // 2
    out[:, :, :half] = Y[:, :, N - half:]
    if cfg.keep_dc:
        out[:, :, half:] = Y[:, :, :nre - half]
    else:
        out[:, :, half:] = Y[:, :, 1:nre - half + 1]
    return out.reshape(n_sf, sf_re)


def capture_cell_search(x, fft_size, M=1):
    """PSS over the three hypotheses (pss.c:446-534), SSS on the symbol before the strongest peak (find_sss.c):
    returns (N_id_2, peak_pos, psr, N_id_1, sf_idx) -- what pbch_file_test / cell_search do with a recorded capture"""
    best = None
    for n2 in range(3):
        pk, pv, psr = pss_find(x, fft_size, n2)
        if best is None or pv > best[2]:
            best = (n2, pk, pv, psr)
    n2, pk, pv, psr = best
    cp = orc().orc_cp_len(fft_size, 144)
    pos = pk - 2 * fft_size - cp
    m0, m1, v0, v1, nid, sf = sss_detect(x[pos:pos + fft_size], fft_size, n2, M)
    return n2, pk, psr, nid, sf


# ------------------------------------------------------------------ soft demodulation / descrambling (orc_modem.c)
LLR_DTYPES = {"s": np.int16, "b": np.int8, "f": np.float32}
QM = {0: 1, 1: 2, 2: 4, 3: 6, 4: 8}


def aligned_empty(n, dtype, align=64):
    """the reference's SSE loads need 16-byte aligned symbol buffers (demod_soft.c:265 _mm_load_ps)"""
    isz = np.dtype(dtype).itemsize
    raw = np.zeros(n * isz + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * isz].view(dtype)


def demod_soft(mod, symbols, kind):
    """orc_demod_soft_{s,b,f}: symbols complex64 [n] -> LLR array of LLR_DTYPES[kind], QM[mod] per symbol"""
    x = np.ascontiguousarray(symbols, np.complex64)
    out = np.zeros(x.size * QM[mod], LLR_DTYPES[kind])
    f = getattr(orc(), "orc_demod_soft_" + kind)
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    assert f(mod, P(x), P(out), x.size) == 0
    return out


def sequence_bits(seed, n):
    c = np.zeros(n, np.uint8)
    f = orc().orc_sequence_bits
    f.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32]
    f(seed, P(c), n)
    return c


def sequence_apply(x, seed):
    kind = {np.dtype(np.int16): "s", np.dtype(np.int8): "c", np.dtype(np.float32): "f"}[x.dtype]
    out = np.zeros_like(x)
    scratch = np.zeros(max(x.size, 1), np.uint8)
    f = getattr(orc(), "orc_sequence_apply_" + kind)
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    f(P(x), P(out), x.size, seed, P(scratch))
    return out


def mod_table(mod):
    """orc_mod_table: the 2^Qm constellation points of srsran_mod_t `mod` as complex64, index = bits MSB first"""
    out = np.zeros(1 << QM[mod], np.complex64)
    f = orc().orc_mod_table
    f.argtypes = [C.c_int, C.c_void_p]
    assert f(mod, P(out)) == out.size
    return out


def modulate_bytes(mod, packed, nbits, seed=0, scramble=False, scaling=1.0):
    """orc_modulate_bytes: byte-packed bits -> [scrambling] -> constellation points x scaling (complex64)"""
    packed = np.ascontiguousarray(packed, np.uint8)
    out = np.zeros(nbits // QM[mod], np.complex64)
    f = orc().orc_modulate_bytes
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_float]
    assert f(mod, P(packed), P(out), nbits, seed, 1 if scramble else 0, scaling) == out.size
    return out


def ulsch_interleaver_lut(nof_sym, Qm, cols):
    """orc_ulsch_interleaver_lut: lut[q position] = g position (36.212 5.2.2.8 without RI / ACK)"""
    lut = np.zeros(nof_sym * Qm, np.uint32)
    f = orc().orc_ulsch_interleaver_lut
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    assert f(nof_sym, Qm, cols, P(lut)) == 0
    return lut


def pusch_seed(rnti, nslot, cell_id):
    f = orc().orc_sequence_pusch_seed
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint16, C.c_uint32, C.c_uint32]
    return f(rnti, nslot, cell_id)


def pdsch_seed(rnti, q, nslot, cell_id):
    f = orc().orc_sequence_pdsch_seed
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint16, C.c_int, C.c_uint32, C.c_uint32]
    return f(rnti, q, nslot, cell_id)


def qam_symbols(mod, n, seed, snr_db=20.0, scale=1.0):
    """random points of the LTE constellations (36.211 7.1, unit average power) + noise; returns complex64 [n]"""
    rng = np.random.default_rng(seed)
    if mod == 0:
        b = rng.integers(0, 2, n)
        s = (1 - 2 * b) * (1 + 1j) / np.sqrt(2)
    else:
        m = 1 << mod  # levels per axis: 2, 4, 8, 16
        lev = 2 * rng.integers(0, m, (2, n)) - (m - 1)
        s = (lev[0] + 1j * lev[1]) / np.sqrt(2 * (m * m - 1) / 3)
    sigma = 10 ** (-snr_db / 20) / np.sqrt(2)
    s = scale * (s + sigma * (rng.normal(size=n) + 1j * rng.normal(size=n)))
    out = aligned_empty(n, np.complex64)
    out[:] = s
    return out


# ------------------------------------------------------------------ NR LDPC rate matching / encoder (orc_ldpc.c)
def ldpc_encode_rm(bg, ls, msg, cdwd_rm_length, fill=7):
    """orc_ldpc_encode_rm: msg uint8 [bgK ls] (254 = filler) -> (bgN - 2) ls bytes; bytes the encoder does not write keep `fill`"""
    g = ldpc_graph(bg, ls)
    out = np.full((g.bgN - 2) * ls, fill, np.uint8)
    f = orc().orc_ldpc_encode_rm
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    assert f(C.byref(g), P(np.ascontiguousarray(msg, np.uint8)), P(out), cdwd_rm_length) == 0
    return out


def ldpc_rm_tx(cw, E, bg, ls, rv, mod, Nref):
    out = np.zeros(E, np.uint8)
    f = orc().orc_ldpc_rm_tx
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    assert f(P(np.ascontiguousarray(cw, np.uint8)), P(out), E, bg, ls, rv, QM[mod], Nref) == 0
    return out


def ldpc_rm_rx(x, base, F, bg, ls, rv, mod, Nref):
    """orc_ldpc_rm_rx: accumulates the soft bits x into a copy of the soft buffer `base` [N]; returns (buffer, n_llr)"""
    typ = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.float32): 2}[x.dtype]
    out = np.array(base, dtype=x.dtype)
    f = orc().orc_ldpc_rm_rx
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    r = f(typ, P(np.ascontiguousarray(x)), P(out), x.size, F, bg, ls, rv, QM[mod], Nref)
    assert r >= 0
    return out, r


def predecoding_single(y, h, scaling, noise_estimate, want_csi=False):
    """orc_predecoding_single (double arithmetic): returns x [, csi]"""
    y, h = np.ascontiguousarray(y, np.complex64), np.ascontiguousarray(h, np.complex64)
    x = np.zeros_like(y)
    csi = np.zeros(y.size, np.float32)
    f = orc().orc_predecoding_single
    f.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_float]
    f(P(y), P(h), P(x), P(csi) if want_csi else None, y.size, scaling, noise_estimate)
    return (x, csi) if want_csi else x


def ldpc_decode_flood(bg, ls, llr, scaling_fctr, max_iter, cdwd_rm_length, crc=None):
    """orc_ldpc_decode_c_flood on one code word: returns (message bits, a-posteriori soft bits [N], return value)"""
    g = ldpc_graph(bg, ls)
    K, N = g.bgK * ls, g.bgN * ls
    llr = np.ascontiguousarray(llr, np.int8)
    out, soft = np.zeros(K, np.uint8), np.zeros(N, np.int8)
    f = orc().orc_ldpc_decode_c_flood
    f.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    poly, order = crc if crc else (0, 0)
    ret = f(C.byref(g), scaling_fctr, max_iter, P(llr), P(out), cdwd_rm_length, poly, order, P(soft))
    return out, soft, ret


# ---------------------------------------------------------------- NR transport-block loop (orc_sch_nr.c)
class NrTbInfo(C.Structure):  # orc_nr_tb_info_t
    _fields_ = [(n, C.c_uint32) for n in ("bg", "Qm", "A", "L_tb", "L_cb", "B", "Bp", "Kp", "Kr", "F", "Z", "G", "Nl", "Nref", "C")]


def sch_nr_tb_info(tbs, R, mod, nof_bits, N_L, Nref):
    cfg = NrTbInfo()
    f = orc().orc_sch_nr_tb_info
    f.argtypes = [C.c_uint32, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    assert f(tbs, R, QM[mod], nof_bits, N_L, Nref, C.byref(cfg)) == 0
    return cfg


def sch_nr_get_E(cfg, j):
    f = orc().orc_sch_nr_get_E
    f.argtypes, f.restype = [C.c_void_p, C.c_uint32], C.c_uint32
    return f(C.byref(cfg), j)


def sch_nr_encode_tb(cfg, rv, payload):
    """orc_sch_nr_encode_tb: payload bytes -> G bits (one per byte)"""
    e = np.zeros(sum(sch_nr_get_E(cfg, r) for r in range(cfg.C)), np.uint8)
    f = orc().orc_sch_nr_encode_tb
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    assert f(C.byref(cfg), rv, P(np.ascontiguousarray(payload, np.uint8)), P(e)) == 0
    return e


def sch_nr_decode_tb(cfg, rv, scaling_fctr, max_iter, llr, softbuf, cb_crc, cb_data):
    """orc_sch_nr_decode_tb: softbuf [C, stride] int8, cb_crc [C] uint8 and cb_data [C, stride] uint8 are updated; returns
    (payload bytes, crc_ok, avg_iter)"""
    out = np.zeros(cfg.A // 8, np.uint8)
    ok, avg = C.c_int(0), C.c_float(0)
    f = orc().orc_sch_nr_decode_tb
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                  C.c_void_p, C.c_void_p]
    llr = np.ascontiguousarray(llr, np.int8)
    assert f(C.byref(cfg), rv, scaling_fctr, max_iter, P(llr) if llr.size else None, P(softbuf), softbuf.shape[1], P(cb_crc), P(cb_data),
             cb_data.shape[1], P(out), C.byref(ok), C.byref(avg)) == 0
    return out, ok.value, avg.value


class RefSchChain:
    """The reference's OWN receive-side FEC chain (oracle/_ref: rm_turbo.c, turbodecoder*.c, crc.c compiled where they lie), driven in the
    order decode_tb_cb drives it (sch.c:389-466: per code block srsran_rm_turbo_rx_lut{,_8bit} into the soft buffer, srsran_tdec_new_cb,
    srsran_tdec_iteration{,_8bit} + srsran_crc_checksum_byte until the CRC matches or max_iterations).  CPU baseline of kind
    "reference" for the transport-block legs and a second check of the oracle's decode_tb restatement.  Needs have_ref()."""

    def __init__(self, llr8=False, max_iterations=10):
        self.ref = C.CDLL(REF_LIB)
        self.llr8, self.max_it = llr8, max_iterations
        self.ref.srsran_rm_turbo_gentables()
        self.tdec = C.create_string_buffer(64 * 1024)
        assert self.ref.srsran_tdec_init(self.tdec, 6144) == 0
        self.crc_tb, self.crc_cb = C.create_string_buffer(4096), C.create_string_buffer(4096)
        assert self.ref.srsran_crc_init(self.crc_tb, 0x1864CFB, 24) == 0 and self.ref.srsran_crc_init(self.crc_cb, 0x1800063, 24) == 0
        self.ref.srsran_crc_checksum_byte.restype = C.c_uint32
        self.ref.srsran_crc_checksum_byte.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.rx = self.ref.srsran_rm_turbo_rx_lut_8bit if llr8 else self.ref.srsran_rm_turbo_rx_lut
        self.rx.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        self.it = self.ref.srsran_tdec_iteration_8bit if llr8 else self.ref.srsran_tdec_iteration
        self.it.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.it.restype = None

    SOFT_STRIDE = 18624  # >= 18600 elements (SRSRAN_HIP_SOFTBUFFER_CB_SIZE) and a multiple of 64 bytes for both LLR widths

    def new_softbuffer(self, n_cb):
        """soft buffers the way the reference allocates them: every code block's row 64-byte aligned (its decoders use aligned loads)"""
        dt = np.int8 if self.llr8 else np.int16
        return aligned_empty(n_cb * self.SOFT_STRIDE, dt).reshape(n_cb, self.SOFT_STRIDE)

    def decode_tb(self, tbs, Qm, rv, e_bits, softbuf, cb_crc):
        """first transmission or HARQ round on `softbuf` (from new_softbuffer, zeroed by the caller for new data) / `cb_crc` [C];
        returns (ok, payload bytes incl. CRC24A, avg half iterations)"""
        seg = cbsegm(tbs)
        Cn, esz = seg["C"], e_bits.dtype.itemsize
        data = np.zeros(tbs // 8 + 16, np.uint8)
        Gp = e_bits.size // Qm
        gamma, n_e = Gp % Cn, Qm * (Gp // Cn)
        noi = 0
        for cb in range(Cn):
            if cb_crc[cb]:
                continue
            K = seg["K1"] if cb < seg["C1"] else seg["K2"]
            rlen = K if Cn == 1 else K - 24
            rp, n_e2 = cb * n_e, n_e
            if cb > Cn - gamma:
                n_e2 = n_e + Qm
                rp = (Cn - gamma) * n_e + (cb - (Cn - gamma)) * n_e2
            assert self.rx(e_bits.ctypes.data + rp * esz, softbuf[cb].ctypes.data, n_e2, orc().orc_tc_cb_index(K), rv) == 0
            assert self.ref.srsran_tdec_new_cb(self.tdec, K) == 0
            out = data.ctypes.data + cb * rlen // 8
            for _ in range(self.max_it):
                self.it(self.tdec, softbuf[cb].ctypes.data, out)
                noi += 1
                if self.ref.srsran_crc_checksum_byte(self.crc_cb if Cn > 1 else self.crc_tb, out, K if Cn > 1 else tbs + 24) == 0:
                    cb_crc[cb] = 1
                    break
        return bool(cb_crc.all()), data, noi / Cn
