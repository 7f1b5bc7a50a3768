"""Parity of the HIP OFDM (de)modulator with the oracle (restated ofdm.c, float64 DFT), through the C ABI.

Tolerance: 1e-4 absolute on unit-scale samples -- the threshold of the reference's own ofdm_test.c:176
(north_star: "within 1e-4 on ... complex IQ").  For un-normalised outputs (scale sqrt(N)) the error
is measured relative to the output RMS."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
TOL = 1e-4

CASES = [
    # prb, symbol_sz, cp_ext, normalize, freq_shift, window_offset, keep_dc
    (6, 0, 0, 1, 0.0, 0.0, 0),       # BASELINE config 1 (1.4 MHz)
    (100, 2048, 0, 1, 0.0, 0.0, 0),  # BASELINE config 2 (20 MHz, standard rate)
    (273, 4096, 0, 1, 0.0, 0.0, 1),  # BASELINE config 3 (NR 100 MHz, keep_dc as ue_dl_nr.c:97-101)
    (100, 0, 0, 0, -0.5, 0.5, 0),    # eNB UL flavour (enb_ul.c:50-57), default 1536-point rate
    (100, 2048, 0, 0, 0.5, 0.0, 0),  # UE UL flavour
    (15, 0, 0, 1, 0.0, 0.0, 0), (25, 0, 0, 1, 0.0, 0.0, 0), (50, 0, 0, 1, 0.0, 0.0, 0), (75, 0, 0, 1, 0.0, 0.0, 0),
    (25, 512, 0, 1, 0.0, 0.0, 0), (50, 1024, 0, 0, 0.0, 0.0, 0), (75, 1536, 0, 1, 0.0, 0.0, 0),
    (200, 3072, 0, 1, 0.0, 0.3, 1),
    (6, 0, 1, 1, 0.0, 0.0, 0), (100, 2048, 1, 1, 0.0, 0.25, 0),  # extended CP
    (110, 2048, 0, 1, 0.0, 1.0, 0),
]


def _rel_err(a, b):
    scale = max(1.0, float(np.sqrt(np.mean(np.abs(b) ** 2))))
    return float(np.abs(a - b).max()) / scale


@pytest.mark.parametrize("case", CASES)
def test_rx_tx_vs_oracle(hiplib, case):
    import srslte_amd as S

    prb, N, cp, norm, fs, wo, kd = case
    cfg = O.ofdm_cfg(prb, N, cp, norm, fs, wo, kd)
    n, nsym, sf_sz, sf_re = O.ofdm_geometry(cfg)
    rng = np.random.default_rng(prb + n)
    n_sf = 5
    re = (rng.uniform(-1, 1, (n_sf, sf_re)) + 1j * rng.uniform(-1, 1, (n_sf, sf_re))).astype(np.complex64)  # ofdm_test.c:152
    tx = S.OfdmBatch(prb, True, N, cp, bool(norm), fs, wo, bool(kd))
    assert (tx.sf_sz, tx.sf_re) == (sf_sz, sf_re)
    t_ref = O.ofdm_tx(cfg, re)
    t_gpu = tx.process(re)
    assert _rel_err(t_gpu, t_ref) < TOL
    x = ((rng.standard_normal((n_sf, sf_sz)) + 1j * rng.standard_normal((n_sf, sf_sz))) * 0.7).astype(np.complex64)
    rx = S.OfdmBatch(prb, False, N, cp, bool(norm), fs, wo, bool(kd))
    assert _rel_err(rx.process(x), O.ofdm_rx(cfg, x)) < TOL


@pytest.mark.parametrize("prb", list(range(6, 111, 13)) + [110])
def test_loopback_all_bandwidths(hiplib, prb):
    """the reference's own test: Tx -> Rx loop-back, MSE < 1e-4 (ofdm_test.c:139-179)"""
    import srslte_amd as S

    rng = np.random.default_rng(prb)
    tx = S.OfdmBatch(prb, True, normalize=True)
    rx = S.OfdmBatch(prb, False, normalize=True)
    re = (rng.uniform(-1, 1, (2, tx.sf_re)) + 1j * rng.uniform(-1, 1, (2, tx.sf_re))).astype(np.complex64)
    back = rx.process(tx.process(re))
    assert np.mean(np.abs(back - re) ** 2) < 1e-4
    assert np.abs(back - re).max() < 1e-4


@pytest.mark.parametrize("prb,N,region,fs", [(6, 0, 2, 0.0), (25, 512, 1, 0.0), (100, 2048, 2, 0.0), (100, 0, 1, 0.0), (50, 1024, 2, 0.5)])
def test_mbsfn_subframes(hiplib, prb, N, region, fs):
    """MBSFN subframe (ofdm.c:424-437,538-555): slot 0 = `region` normal-CP symbols, a guard gap, then extended-CP
    symbols; slot 1 a regular extended-CP slot.  Batch API against the oracle, plus the pmch_test.c loop-back."""
    import srslte_amd as S
    from srslte_amd import capi

    cfg = O.ofdm_cfg(prb, N, 1, 1, fs, 0.0, 0, mbsfn_region=region)
    n, nsym, sf_sz, sf_re = O.ofdm_geometry(cfg)
    rng = np.random.default_rng(prb + region)
    n_sf = 3
    re = (rng.uniform(-1, 1, (n_sf, sf_re)) + 1j * rng.uniform(-1, 1, (n_sf, sf_re))).astype(np.complex64)
    tx = S.OfdmBatch(prb, True, N, capi.CP_EXT, True, fs, mbsfn_region=region)
    rx = S.OfdmBatch(prb, False, N, capi.CP_EXT, True, fs, mbsfn_region=region)
    t_ref, t_gpu = O.ofdm_tx(cfg, re), tx.process(re)
    assert _rel_err(t_gpu, t_ref) < TOL
    # the layout differs from a plain extended-CP subframe (else this test would prove nothing)
    assert _rel_err(t_gpu, O.ofdm_tx(O.ofdm_cfg(prb, N, 1, 1, fs, 0.0, 0), re)) > 0.1
    x = ((rng.standard_normal((n_sf, sf_sz)) + 1j * rng.standard_normal((n_sf, sf_sz))) * 0.7).astype(np.complex64)
    assert _rel_err(rx.process(x), O.ofdm_rx(cfg, x)) < TOL
    if fs == 0.0:
        assert np.abs(rx.process(t_gpu) - re).max() < 1e-4
    # the other region on the same objects (srsran_ofdm_set_non_mbsfn_region at run time, enb_dl.c:346)
    other = 3 - region
    S.lib().srsran_hip_ofdm_batch_set_non_mbsfn_region(tx._h, other)
    cfg2 = O.ofdm_cfg(prb, N, 1, 1, fs, 0.0, 0, mbsfn_region=other)
    assert _rel_err(tx.process(re), O.ofdm_tx(cfg2, re)) < TOL
    plain = S.OfdmBatch(prb, True, N, capi.CP_EXT, True, fs)
    assert S.lib().srsran_hip_ofdm_batch_set_non_mbsfn_region(plain._h, 1) == capi.SRSRAN_ERROR_INVALID_INPUTS


def test_mbsfn_handle_api(hiplib):
    """srsran_ofdm_{tx,rx}_init_mbsfn + set_non_mbsfn_region + set_normalize, as pmch_test.c:207-225,316,347 uses them"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(8)
    for prb, region in ((100, 2), (25, 1)):
        ocfg = O.ofdm_cfg(prb, 0, 1, 1, 0.0, 0.0, 0, mbsfn_region=region)
        n, nsym, sf_sz, sf_re = O.ofdm_geometry(ocfg)
        re_in, t_out = np.zeros(sf_re, np.complex64), np.full(sf_sz, 7 + 7j, np.complex64)
        t_in, re_out = np.zeros(sf_sz, np.complex64), np.zeros(sf_re, np.complex64)
        ifft, fft = capi.Ofdm(), capi.Ofdm()
        assert lib.srsran_ofdm_tx_init_mbsfn(C.byref(ifft), capi.CP_EXT, O.P(re_in), O.P(t_out), prb) == 0
        lib.srsran_ofdm_set_non_mbsfn_region(C.byref(ifft), region)
        lib.srsran_ofdm_set_normalize(C.byref(ifft), True)
        assert lib.srsran_ofdm_rx_init_mbsfn(C.byref(fft), capi.CP_EXT, O.P(t_in), O.P(re_out), prb) == 0
        lib.srsran_ofdm_set_non_mbsfn_region(C.byref(fft), region)
        lib.srsran_ofdm_set_normalize(C.byref(fft), True)
        assert ifft.mbsfn_subframe and ifft.non_mbsfn_region == region and ifft.nof_symbols == 6
        re = (rng.uniform(-1, 1, sf_re) + 1j * rng.uniform(-1, 1, sf_re)).astype(np.complex64)
        re_in[:] = re
        lib.srsran_ofdm_tx_sf(C.byref(ifft))
        ref = O.ofdm_tx(ocfg, re[None])[0]
        written = np.abs(t_out - (7 + 7j)) > 0
        # the gap between the two regions is left untouched in the caller's buffer (ofdm.c:551-553)
        ext, n0, n1 = O.orc().orc_cp_len(n, 512), O.orc().orc_cp_len(n, 160), O.orc().orc_cp_len(n, 144)
        gap = ext - n0 if region == 1 else 2 * ext - n0 - n1
        assert (~written).sum() == gap
        assert np.abs(t_out[written] - ref[written]).max() < TOL
        t_in[:] = np.where(written, t_out, 0)
        lib.srsran_ofdm_rx_sf(C.byref(fft))
        assert np.abs(re_out - re).max() < 1e-4
        lib.srsran_ofdm_tx_free(C.byref(ifft))
        lib.srsran_ofdm_rx_free(C.byref(fft))


def test_handle_api(hiplib):
    """srsran_ofdm_{rx,tx}_init_cfg / _sf / set_prb / free on host buffers, incl. the reference's side effects"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(3)
    for prb, N, norm, fs, wo in ((100, 2048, True, 0.0, 0.0), (25, 0, False, -0.5, 0.5), (6, 0, True, 0.0, 0.0)):
        ocfg = O.ofdm_cfg(prb, N, 0, int(norm), fs, wo, 0)
        n, nsym, sf_sz, sf_re = O.ofdm_geometry(ocfg)
        tin = np.ones(sf_sz, np.complex64)
        rout = np.zeros(sf_re, np.complex64)
        q = capi.Ofdm()
        cfg = capi.OfdmCfg()
        cfg.nof_prb, cfg.in_buffer, cfg.out_buffer, cfg.cp = prb, tin.ctypes.data, rout.ctypes.data, capi.CP_NORM
        cfg.normalize, cfg.freq_shift_f, cfg.rx_window_offset, cfg.symbol_sz = norm, fs, wo, N
        assert lib.srsran_ofdm_rx_init_cfg(C.byref(q), C.byref(cfg)) == 0
        assert (q.cfg.symbol_sz, q.nof_re, q.sf_sz, q.slot_sz, q.nof_symbols) == (n, 12 * prb, sf_sz, sf_sz // 2, 7)
        assert not tin.any(), "init must zero the input buffer (ofdm.c:142-147)"
        x = ((rng.standard_normal(sf_sz) + 1j * rng.standard_normal(sf_sz)) * 0.7).astype(np.complex64)
        tin[:] = x
        lib.srsran_ofdm_rx_sf(C.byref(q))
        ref = O.ofdm_rx(ocfg, x[None])[0]
        assert _rel_err(rout, ref) < TOL
        if fs:
            assert not np.array_equal(tin, x), "the freq-shifted input is written back (ofdm.c:455-457)"
        else:
            assert np.array_equal(tin, x)
        lib.srsran_ofdm_rx_free(C.byref(q))
        assert q.max_prb == 0 and not q.tmp
        # tx
        rin = np.ones(sf_re, np.complex64)
        tout = np.zeros(sf_sz, np.complex64)
        q = capi.Ofdm()
        cfg.in_buffer, cfg.out_buffer = rin.ctypes.data, tout.ctypes.data
        assert lib.srsran_ofdm_tx_init_cfg(C.byref(q), C.byref(cfg)) == 0
        re = (rng.uniform(-1, 1, sf_re) + 1j * rng.uniform(-1, 1, sf_re)).astype(np.complex64)
        rin[:] = re
        lib.srsran_ofdm_tx_sf(C.byref(q))
        assert _rel_err(tout, O.ofdm_tx(ocfg, re[None])[0]) < TOL
        lib.srsran_ofdm_tx_free(C.byref(q))
    # srsran_ofdm_rx_init + set_prb to a smaller bandwidth
    tin = np.zeros(15 * 1536, np.complex64)
    rout = np.zeros(14 * 1200, np.complex64)
    q = capi.Ofdm()
    assert lib.srsran_ofdm_rx_init(C.byref(q), capi.CP_NORM, tin.ctypes.data, rout.ctypes.data, 100) == 0
    assert q.cfg.symbol_sz == 1536 and lib.srsran_symbol_sz(100) == 1536 and lib.srsran_symbol_sz_power2(100) == 2048
    assert lib.srsran_ofdm_rx_set_prb(C.byref(q), capi.CP_NORM, 25) == 0
    assert (q.cfg.symbol_sz, q.nof_re, q.sf_sz) == (384, 300, 5760)
    x = ((rng.standard_normal(5760) + 1j * rng.standard_normal(5760)) * 0.7).astype(np.complex64)
    tin[:5760] = x
    lib.srsran_ofdm_rx_sf(C.byref(q))
    assert _rel_err(rout[:14 * 300], O.ofdm_rx(O.ofdm_cfg(25), x[None])[0]) < TOL
    lib.srsran_ofdm_rx_free(C.byref(q))
    assert lib.srsran_symbol_sz(0) == -1 and lib.srsran_symbol_sz(111) == -1


def test_full_size_linearity_property(hiplib):
    """BASELINE size (thousands of 20 MHz subframes): the demodulator is linear and batch-invariant"""
    import srslte_amd as S

    rng = np.random.default_rng(1)
    rx = S.OfdmBatch(100, False, 2048, normalize=True)
    n_sf = 1024
    a = ((rng.standard_normal((4, rx.sf_sz)) + 1j * rng.standard_normal((4, rx.sf_sz))) * 0.5).astype(np.complex64)
    big = np.tile(a, (n_sf // 4, 1))
    big[1::4] = a[0] + a[2]  # replace every second row of the pattern by a sum
    out = rx.process(big)
    assert np.array_equal(out[0::4], np.tile(out[0], (n_sf // 4, 1)))
    assert np.abs(out[1] - (out[0] + out[2])).max() < 1e-4
    ref = O.ofdm_rx(O.ofdm_cfg(100, 2048, 0, 1), a[:1])
    assert np.abs(out[4 * 100] - ref[0]).max() < 1e-4
