"""srsran_sync_t and the helpers it is made of (srsran_cfo_t, srsran_cp_synch_t, PSS filter / channel estimate /
CFO / decimated and sliding-window search), through the C ABI, against the oracle and against ground truth.

Component results are compared with the restated reference (float, 1e-4 relative unless noted); srsran_sync_find
is checked the way the reference's sync_test.c:130-176 checks it -- peak position, subframe index, CP, cell id --
for every stage combination (integer / CP / PSS CFO stages, known N_id_1, TDD, extended CP, thresholds)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _noise(rng, n, amp):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * amp).astype(np.complex64)


def _relerr(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def test_cfo_correct_vs_oracle(hiplib):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(1)
    n = 1920
    x = _noise(rng, n, 1.0)
    h = capi.Cfo()
    assert lib.srsran_cfo_init(C.byref(h), n) == 0
    out = np.zeros(n, np.complex64)
    for freq in (0.0, 1.3e-4, -7.7e-4, 0.013, -0.2):
        lib.srsran_cfo_correct(C.byref(h), O.P(x), O.P(out), freq)
        assert _relerr(out, O.cfo_correct(x, freq)) < 1e-6, freq
    # the float oscillators drift from the ideal exponential exactly like the reference's (vector_simd.c:1692-1739)
    lib.srsran_cfo_correct(C.byref(h), O.P(x), O.P(out), 0.013)
    ideal = x * np.exp(2j * np.pi * 0.013 * np.arange(n))
    assert 1e-7 < _relerr(out, ideal) < 1e-4
    # srsran_cfo_correct_offset is table driven (cfo.c:120-134): phase quantised to 1/4096 of a turn
    lib.srsran_cfo_correct_offset(C.byref(h), O.P(x), O.P(out), -0.2, 100, 500)
    tab = np.zeros(n, np.complex64)
    O.orc().orc_cexptab_gen.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    O.orc().orc_cexptab_gen(O.P(tab), -0.2, n)
    assert _relerr(out[:500], tab[100:600] * x[:500]) < 1e-6
    # SRSRAN_CFO_USE_EXP_TABLE 0: resize is accepted and changes nothing (cfo.c:83-95)
    assert lib.srsran_cfo_resize(C.byref(h), 960) == 0 and h.nsamples == n
    lib.srsran_cfo_free(C.byref(h))
    assert not h.cur_cexp


@pytest.mark.parametrize("N,nsym,max_off", [(128, 3, 128), (512, 7, 200), (2048, 2, 4000)])
def test_cp_synch_vs_oracle(hiplib, N, nsym, max_off):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(N)
    cp = O.orc().orc_cp_len(N, 144)
    # OFDM-like signal: symbols with a true cyclic prefix, starting 37 samples into the buffer, rotated by a CFO
    syms = []
    for i in range(nsym + 2):
        body = _noise(rng, N, 1.0)
        cl = cp if i % 7 else cp + 1
        syms.append(np.concatenate([body[-cl:], body]))
    sig = np.concatenate([_noise(rng, 37, 0.1)] + syms + [_noise(rng, 2 * N, 0.1)])
    sig = (sig * np.exp(2j * np.pi * 0.21 * np.arange(sig.size) / N)).astype(np.complex64)
    q = capi.CpSynch()
    assert lib.srsran_cp_synch_init(C.byref(q), N) == 0
    idx = lib.srsran_cp_synch(C.byref(q), O.P(sig), max_off, nsym, cp)
    ref_idx, ref_corr = O.cp_synch(sig, N, max_off, nsym, cp)
    got = np.ctypeslib.as_array(C.cast(q.corr, C.POINTER(C.c_float)), shape=(2 * min(max_off, N),)).view(np.complex64)
    assert idx == ref_idx == 37
    assert _relerr(got, ref_corr) < 1e-4
    # the phase of the peak is the CFO (sync.c:576-590): -arg / 2 pi = 0.21 sub-carriers
    assert abs(-np.angle(got[idx]) / (2 * np.pi) - 0.21) < 0.01
    lib.srsran_cp_synch_free(C.byref(q))


@pytest.mark.parametrize("N", [128, 384, 1024, 2048])
def test_pss_filter_chest_cfo(hiplib, N):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(N)
    prb = {128: 6, 384: 25, 1024: 75, 2048: 100}[N]
    for n2 in range(3):
        sig, end = O.cell_signal(150 + n2, prb, N)
        pss_sym = sig[end - N:end] + _noise(rng, N, 0.02)
        q = capi.Pss()
        assert lib.srsran_pss_init_fft(C.byref(q), 15 * N, N) == 0
        assert lib.srsran_pss_set_N_id_2(C.byref(q), n2) == 0
        out = np.zeros(N, np.complex64)
        wide = (pss_sym + _noise(rng, N, 0.5)).astype(np.complex64)  # out-of-band energy the filter must remove
        q.chest_on_filter = True
        lib.srsran_pss_filter(C.byref(q), O.P(wide), O.P(out))
        ref, ref_ce = O.pss_filter(wide, N, n2, want_ce=True)
        assert _relerr(out, ref) < 1e-4
        ce = np.array(q.tmp_ce[:], np.float32).view(np.complex64)
        assert _relerr(ce, ref_ce) < 1e-4
        ce2 = np.zeros(62, np.complex64)
        assert lib.srsran_pss_chest(C.byref(q), O.P(wide), O.P(ce2)) == 0
        assert _relerr(ce2, ref_ce) < 1e-4
        # CFO of eps sub-carriers: estimate from the two halves of the PSS symbol (pss.c:611-640)
        for eps in (0.0, 0.12, -0.3):
            rot = (pss_sym * np.exp(2j * np.pi * eps * np.arange(N) / N)).astype(np.complex64)
            est = lib.srsran_pss_cfo_compute(C.byref(q), O.P(rot))
            assert abs(est - O.pss_cfo_compute(rot, N, n2)) < 1e-4
            assert abs(est - eps) < 0.03
        # interference cancellation removes the PSS from its symbol (pss.c:536-558)
        clean = pss_sym.copy()
        lib.srsran_pss_filter(C.byref(q), O.P(clean), O.P(out))
        lib.srsran_pss_sic(C.byref(q), O.P(clean))
        assert np.mean(np.abs(clean) ** 2) < 0.05 * np.mean(np.abs(pss_sym) ** 2)
        lib.srsran_pss_free(C.byref(q))


@pytest.mark.parametrize("N,win", [(128, 60), (512, 100), (2048, 333)])
def test_pss_sliding_window_mode(hiplib, N, win):
    """frame_size < fft_size: the tracking configuration of srsran_sync_t (max_offset of a few dozen samples)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(win)
    prb = {128: 6, 512: 25, 2048: 100}[N]
    sig, end = O.cell_signal(7, prb, N)
    start = end - N - win // 3  # PSS symbol begins win/3 samples into the window
    x = (sig[start:start + win + N] + _noise(rng, win + N, 0.05)).astype(np.complex64)
    q = capi.Pss()
    assert lib.srsran_pss_init_fft(C.byref(q), win, N) == 0
    lib.srsran_pss_set_ema_alpha(C.byref(q), 1.0)
    for n2 in range(3):
        lib.srsran_pss_set_N_id_2(C.byref(q), n2)
        pv = C.c_float()
        pos = lib.srsran_pss_find_pss(C.byref(q), O.P(x), C.byref(pv))
        rpos, rpeak, rpsr = O.pss_find(x, N, n2, frame_size=win)
        assert pos == rpos
        assert abs(pv.value - rpsr) <= 1e-3 * rpsr and abs(q.peak_value - rpeak) <= 1e-4 * rpeak
        if n2 == 7 % 3:
            assert pos == win // 3 + N
    lib.srsran_pss_free(C.byref(q))


@pytest.mark.parametrize("N,dec", [(512, 2), (2048, 4), (1536, 3)])
def test_pss_decimated_search(hiplib, N, dec):
    """srsran_pss_init_fft_offset_decim: keep every dec-th sample behind a 4-tap low-pass (filter.c), search at the
    reduced rate, scale the position back (pss.c:460-470,521-525)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(dec)
    prb = {512: 25, 2048: 100, 1536: 100}[N]
    frame = 15 * N
    sig, end = O.cell_signal(301, prb, N)
    delay = 5 * dec * 7
    x = _noise(rng, frame, 0.02)
    x[delay:] += sig[:frame - delay]
    q = capi.Pss()
    assert lib.srsran_pss_init_fft_offset_decim(C.byref(q), frame, N, 0, dec) == 0
    assert (q.fft_size, q.frame_size, q.filter.num_taps) == (N // dec, frame // dec, 4)
    lib.srsran_pss_set_ema_alpha(C.byref(q), 1.0)
    lib.srsran_pss_set_N_id_2(C.byref(q), 301 % 3)
    pv = C.c_float()
    pos = lib.srsran_pss_find_pss(C.byref(q), O.P(x), C.byref(pv))
    taps = np.ctypeslib.as_array(C.cast(q.filter.taps, C.POINTER(C.c_float)), shape=(4,)).astype(np.float64)
    ds = np.concatenate([np.zeros(3, np.complex128), x[::dec][:frame // dec].astype(np.complex128)])
    y = sum(taps[t] * ds[t:t + frame // dec] for t in range(4)).astype(np.complex64)
    rpos, rpeak, rpsr = O.pss_find(y, N // dec, 301 % 3)
    assert pos == (rpos - 2) * dec
    assert abs(q.peak_value - rpeak) <= 1e-4 * rpeak and abs(pv.value - rpsr) <= 1e-3 * rpsr
    assert abs(pos - (delay + end)) <= 2 * dec  # the filter delay is only compensated to within the decimation grid
    lib.srsran_pss_free(C.byref(q))


def _sync(lib, capi, frame, max_off, N, **kw):
    q = capi.Sync()
    assert lib.srsran_sync_init(C.byref(q), frame, max_off, N) == 0
    lib.srsran_sync_set_threshold(C.byref(q), kw.get("threshold", 2.0))
    lib.srsran_sync_set_em_alpha(C.byref(q), 1.0)
    lib.srsran_sync_set_sss_algorithm(C.byref(q), kw.get("alg", capi.SSS_PARTIAL_3))
    return q


@pytest.mark.parametrize("prb,N", [(6, 128), (25, 384), (50, 768), (100, 1536)])
def test_sync_find_like_sync_test(hiplib, prb, N):
    """sync_test.c:130-176: every N_id_2, subframe 0 and 5, a sample offset; asserts position, subframe and CP"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    flen = 15 * N
    for cp_ext in (False, True):
        q = _sync(lib, capi, flen, flen, N, threshold=3.0)  # the reference's 5.0 holds for its cells 0..49 only (503: PSR 4.15)
        lib.srsran_sync_set_cp(C.byref(q), capi.CP_EXT if cp_ext else capi.CP_NORM)
        for cid in (0, 1, 2, 49, 334, 503):
            lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
            for sf5 in (False, True):
                sig, end = O.cell_signal(cid, prb, N, cp_ext, sf5=sf5, n_sf=1)
                offset = 100 + cid % 17
                buf = np.zeros(2 * flen, np.complex64)
                buf[offset:offset + flen] = sig
                pk = C.c_uint32()
                ret = lib.srsran_sync_find(C.byref(q), O.P(buf), 0, C.byref(pk))
                assert ret == capi.SYNC_FOUND
                assert pk.value == offset + end
                assert lib.srsran_sync_get_sf_idx(C.byref(q)) == (5 if sf5 else 0)
                assert lib.srsran_sync_get_cp(C.byref(q)) == (capi.CP_EXT if cp_ext else capi.CP_NORM)
                assert lib.srsran_sync_get_cell_id(C.byref(q)) == cid
                assert lib.srsran_sync_sss_detected(C.byref(q)) and lib.srsran_sync_sss_available(C.byref(q))
                assert q.frame_type == capi.FDD
        lib.srsran_sync_free(C.byref(q))


def test_sync_find_stages(hiplib):
    """CP-based and PSS-based CFO stages, integer CFO trial, known N_id_1, TDD detection, thresholds, no-space"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(12)
    prb, N = 25, 384
    flen = 15 * N
    cid = 236
    sig, end = O.cell_signal(cid, prb, N, n_sf=2)
    # a data-like subframe around the sync signals so that the CP correlation has something to work with
    cfg = O.ofdm_cfg(prb, N, 0, 1)
    n, nsym, sf_sz, sf_re = O.ofdm_geometry(cfg)
    k = 12 * prb // 2 - 36
    grid = np.ones((2, nsym, 12 * prb), np.float32)
    grid[0, 5:7, k:k + 72] = 0  # keep the 72 central carriers of the PSS / SSS symbols free of data
    data = O.ofdm_tx(cfg, (grid.reshape(2, -1) * (rng.choice([-0.7, 0.7], (2, sf_re)) + 1j * rng.choice([-0.7, 0.7], (2, sf_re)))).astype(np.complex64)).reshape(-1)
    tx = (sig + data).astype(np.complex64)
    offset = 211

    def frame(cfo):
        buf = _noise(rng, 2 * flen + 4 * N, 0.01)
        buf[offset:offset + 2 * flen] += tx
        return (buf * np.exp(2j * np.pi * cfo * np.arange(buf.size) / N)).astype(np.complex64)

    # 1. fractional CFO: CP stage then PSS stage recover it (cfo in sub-carriers = 15 kHz units)
    q = _sync(lib, capi, flen, flen, N)
    lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
    lib.srsran_sync_set_cfo_cp_enable(C.byref(q), True, 7)
    lib.srsran_sync_set_cfo_pss_enable(C.byref(q), True)
    lib.srsran_sync_set_pss_filt_enable(C.byref(q), True)
    pk = C.c_uint32()
    ret = lib.srsran_sync_find(C.byref(q), O.P(frame(0.3)), 0, C.byref(pk))
    assert ret == capi.SYNC_FOUND and pk.value == offset + end
    assert abs(lib.srsran_sync_get_cfo(C.byref(q)) - 0.3) < 0.03
    assert q.cfo_cp_is_set and q.cfo_pss_is_set and abs(q.cfo_cp_mean - 0.3) < 0.05
    assert lib.srsran_sync_get_cell_id(C.byref(q)) == cid and lib.srsran_sync_get_sf_idx(C.byref(q)) == 0
    # second frame: exponential averages move by cfo_ema_alpha (sync.c:686-690)
    before = q.cfo_cp_mean
    lib.srsran_sync_find(C.byref(q), O.P(frame(0.2)), 0, C.byref(pk))
    assert before - 0.1 * 0.2 < q.cfo_cp_mean < before
    lib.srsran_sync_cfo_reset(C.byref(q), 1500.0)
    assert abs(lib.srsran_sync_get_cfo(C.byref(q)) - 0.1) < 1e-6 and not q.cfo_cp_is_set
    lib.srsran_sync_free(C.byref(q))

    # 2. integer CFO trial (sync.c:592-622).  The replica built with bin offset s is conj(p[n] e^{+j 2 pi s n / N}) and is
    # used in a CONVOLUTION, so it matches a signal shifted by -s sub-carriers: the reference reports cfo_i_value = -CFO
    # (and its correction then doubles the offset; the stage is off by default in ue_sync.c).  Same behaviour here.
    for icfo in (1, -1, 0):
        q = _sync(lib, capi, flen, flen, N)
        lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
        lib.srsran_sync_set_cfo_i_enable(C.byref(q), True)
        assert q.cfo_i_initiated and q.pss_i[1].fft_size == N
        ret = lib.srsran_sync_find(C.byref(q), O.P(frame(float(icfo))), 0, C.byref(pk))
        assert ret == capi.SYNC_FOUND and q.cfo_i_value == -icfo and pk.value == offset + end
        want = q.pss_i[1] if q.cfo_i_value > 0 else (q.pss_i[0] if q.cfo_i_value < 0 else q.pss)
        assert C.addressof(lib.srsran_sync_get_cur_pss_obj(C.byref(q)).contents) == C.addressof(want)
        assert abs(lib.srsran_sync_get_cfo(C.byref(q)) - q.cfo_i_value) < 1e-6
        if icfo == 0:
            assert lib.srsran_sync_get_cell_id(C.byref(q)) == cid
        lib.srsran_sync_free(C.byref(q))

    # 3. known N_id_1: only the subframe is decided, by correlating with both stored SSS (sync.c:519-548)
    for sf5 in (False, True):
        s5, e5 = O.cell_signal(cid, prb, N, sf5=sf5, n_sf=1)
        buf = _noise(rng, 2 * flen, 0.01)
        buf[offset:offset + flen] += s5
        q = _sync(lib, capi, flen, flen, N)
        lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
        assert lib.srsran_sync_set_N_id_1(C.byref(q), cid // 3) == 0 and q.sss_generated
        assert lib.srsran_sync_set_N_id_1(C.byref(q), 168) == capi.SRSRAN_ERROR_INVALID_INPUTS
        lib.srsran_sync_set_frame_type(C.byref(q), capi.FDD)
        assert lib.srsran_sync_find(C.byref(q), O.P(buf), 0, C.byref(pk)) == capi.SYNC_FOUND
        assert lib.srsran_sync_get_sf_idx(C.byref(q)) == (5 if sf5 else 0) and lib.srsran_sync_sss_detected(C.byref(q))
        assert lib.srsran_sync_sss_correlation_peak(C.byref(q)) > 1.2
        lib.srsran_sync_free(C.byref(q))

    # 4. TDD: SSS three symbols before the PSS; frame type detected from the stronger SSS correlation
    st, et = O.cell_signal(cid, prb, N, tdd=True, n_sf=2)
    buf = _noise(rng, 2 * flen + 4 * N, 0.01)
    buf[offset:offset + 2 * flen] += st
    q = _sync(lib, capi, 2 * flen, 2 * flen, N, alg=capi.SSS_FULL)
    lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
    assert lib.srsran_sync_find(C.byref(q), O.P(buf), 0, C.byref(pk)) == capi.SYNC_FOUND
    assert pk.value == offset + et and q.frame_type == capi.TDD
    assert lib.srsran_sync_get_cell_id(C.byref(q)) == cid and lib.srsran_sync_get_sf_idx(C.byref(q)) == 1
    lib.srsran_sync_free(C.byref(q))

    # 5. thresholds and space checks
    q = _sync(lib, capi, flen, flen, N, threshold=1e6)
    assert lib.srsran_sync_find(C.byref(q), O.P(frame(0.0)), 0, C.byref(pk)) == capi.SYNC_ERROR  # N_id_2 not set
    lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
    assert lib.srsran_sync_find(C.byref(q), O.P(frame(0.0)), 0, C.byref(pk)) == capi.SYNC_NOFOUND
    assert pk.value == offset + end
    lib.srsran_sync_set_threshold(C.byref(q), 2.0)
    early = np.zeros(2 * flen, np.complex64)
    early[:flen - (end - N - 10)] = sig[end - N - 10:flen]  # PSS symbol right at the start: no room for the SSS
    assert lib.srsran_sync_find(C.byref(q), O.P(early), 0, C.byref(pk)) == capi.SYNC_FOUND_NOSPACE
    assert pk.value == N + 10
    assert lib.srsran_sync_resize(C.byref(q), flen + 1, flen, N) == capi.SRSRAN_ERROR
    assert lib.srsran_sync_resize(C.byref(q), flen, 1000, 100) == capi.SRSRAN_ERROR_INVALID_INPUTS
    # tracking configuration: search +-30 samples around the expected position (sliding dot-product PSS mode)
    assert lib.srsran_sync_resize(C.byref(q), flen, 60, N) == 0
    lib.srsran_sync_set_N_id_2(C.byref(q), cid % 3)
    lib.srsran_sync_set_em_alpha(C.byref(q), 1.0)
    fo = offset + end - N - 30
    assert lib.srsran_sync_find(C.byref(q), O.P(frame(0.0)), fo, C.byref(pk)) == capi.SYNC_FOUND
    assert fo + pk.value == offset + end and lib.srsran_sync_get_cell_id(C.byref(q)) == cid
    lib.srsran_sync_free(C.byref(q))


def test_detect_cp_vs_oracle(hiplib):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(3)
    prb, N = 50, 768
    for cp_ext in (False, True):
        cfg = O.ofdm_cfg(prb, N, 1 if cp_ext else 0, 1)
        n, nsym, sf_sz, sf_re = O.ofdm_geometry(cfg)
        x = O.ofdm_tx(cfg, (rng.uniform(-1, 1, (1, sf_re)) + 1j * rng.uniform(-1, 1, (1, sf_re))).astype(np.complex64))[0]
        x = (x + _noise(rng, sf_sz, 0.05)).astype(np.complex64)
        q = _sync(lib, capi, sf_sz, sf_sz, N)
        peak = sf_sz // 2
        got = lib.srsran_sync_detect_cp(C.byref(q), O.P(x), peak)
        ref, m = O.detect_cp(x, peak, N)
        assert got == ref == (1 if cp_ext else 0)
        assert abs(q.M_norm_avg - m[0]) < 1e-5 and abs(q.M_ext_avg - m[1]) < 1e-5
        lib.srsran_sync_free(C.byref(q))
