"""Parity of the HIP turbo decoder with the oracle (the restated reference), through the C ABI.

Bar: bit-exact hard decisions AND bit-exact decision LLRs, for the decoder the reference's AUTO mode
selects for each K (turbodecoder.c:381-408), for every number of half iterations, at error-free,
waterfall and hopeless SNR (the reference's implementations only agree with each other on blocks
that decode; we match the specific implementation, so failing blocks match too)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["throughput kernel", "latency kernel", "latency kernel, two waves"])
def tdec_kernel(request, hiplib):
    """every test of this module runs twice: with the throughput kernel (8 code blocks per wave, turbo_kernels.hip) and with the latency kernel
    (one code block per wave, states across lanes, turbo_lat_kernels.hip) wherever the latter exists (16 sub-blocks, and 8 sub-blocks 16-bit) --
    SRSRAN_HIP_TDEC_LAT = 0 / 1; unset, the library picks by batch size (turbo_device.h: kLatMaxBlocks)"""
    # (third run: the two-wave form of the latency kernel -- forward and backward recursion of a block at once, 16 sub-blocks / 16-bit -- SRSRAN_HIP_TDEC_LAT2;
    # the scalar decoder's latency kernel, turbo_gen_lat_kernels.hip, follows SRSRAN_HIP_TDEC_LAT as well)
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", b"0" if request.param == "throughput kernel" else b"1") == 0
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT2", b"1" if request.param == "latency kernel, two waves" else b"0") == 0
    yield request.param
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", None) == 0
    assert hiplib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT2", None) == 0


def _check(S, K, impl_g, impl_o, n_cb, snr, nits, seed, sb_layout=0):
    msgs, llr = O.turbo_llrs(K, n_cb, snr, seed)
    src = llr
    if sb_layout:
        nb = 16 if (K % 16 == 0 and K > 800) else 8
        src = np.stack([O.natural_to_sb_layout(llr[i], K, nb) for i in range(n_cb)])
    dec = S.TdecBatch(K, n_cb, impl_g)
    for nit in nits:
        ref, ref_llr = O.turbo_decode(src, nit, K, impl_o, sb_layout, want_llr=True)
        out, out_llr = dec.decode(src, nit, sb_layout, want_llr=True)
        assert np.array_equal(ref, out), "K=%d nit=%d snr=%g: %d code blocks differ" % (K, nit, snr, np.any(ref != out, axis=1).sum())
        assert np.array_equal(ref_llr, out_llr), "K=%d nit=%d snr=%g: decision LLRs differ" % (K, nit, snr)
    dec.free()


@pytest.mark.parametrize("K", [40, 48, 208, 400, 408, 504, 512, 800, 816, 1008, 1024, 1120, 2048, 2112, 4160, 6144])
def test_auto_all_regimes(hiplib, K):
    import srslte_amd as S
    from srslte_amd import capi

    for snr in (3.0, -1.0, -4.0):
        _check(S, K, capi.TDEC_AUTO, O.ORC_TDEC_AUTO, 11, snr, (1, 2, 3, 4, 7, 8), seed=K * 7 + int(snr * 3))


def test_every_block_size_once(hiplib):
    """all 188 LTE block sizes, 3 blocks each, 4 half iterations"""
    import srslte_amd as S
    from srslte_amd import capi

    for i, K in enumerate(O.tc_sizes()):
        _check(S, K, capi.TDEC_AUTO, O.ORC_TDEC_AUTO, 3, -1.0 if i % 2 else 2.0, (4,), seed=K)


@pytest.mark.parametrize("impl_g,impl_o,K", [("GENERIC", O.ORC_TDEC_GENERIC, 1024), ("SSE_WINDOW", O.ORC_TDEC_SSE_WINDOW, 6144),
                                              ("AVX_WINDOW", O.ORC_TDEC_AVX_WINDOW, 1024), ("SSE_WINDOW", O.ORC_TDEC_SSE_WINDOW, 328)])
def test_manual_implementations(hiplib, impl_g, impl_o, K):
    import srslte_amd as S
    from srslte_amd import capi

    _check(S, K, getattr(capi, "TDEC_" + impl_g), impl_o, 5, -1.0, (1, 2, 5, 8), seed=K)


@pytest.mark.parametrize("K", [6144, 5824, 2112, 1024, 1008, 512])
def test_rm_turbo_subblock_layout(hiplib, K):
    """input in the layout srsran_rm_turbo_rx_lut produces for the window decoders (turbodecoder_iter.h:88-102)"""
    import srslte_amd as S
    from srslte_amd import capi

    _check(S, K, capi.TDEC_AUTO, O.ORC_TDEC_AUTO, 9, -1.0, (1, 2, 8), seed=K + 1, sb_layout=1)


def _check8(S, K, impl_g, impl_o, n_cb, snr, nits, seed, scale=12.0, sb_layout=0, via16=False):
    """8-bit LLR API (srsran_tdec_run_all_8bit); via16: a manual 8-bit decoder driven through the int16 entry point,
    which truncates the LLRs to int8 (convert_16_to_8, turbodecoder.c:449-453) -- the only way the reference can run
    its manually selected 8-bit decoders (its 8-bit entry point leaves the interleaver index unset in manual mode)"""
    msgs, llr = O.turbo_llrs_8bit(K, n_cb, snr, seed, scale)
    src = llr
    if sb_layout:
        nb = S.lib().srsran_tdec_autoimp_get_subblocks_8bit(K)
        src = np.stack([O.natural_to_sb_layout(llr[i], K, nb) for i in range(n_cb)])
    dec = S.TdecBatch(K, n_cb, impl_g, llr8=True)
    for nit in nits:
        ref, ref_llr = O.turbo_decode_8bit(src, nit, K, impl_o, sb_layout, want_llr=True)
        out, out_llr = dec.decode(src.astype(np.int16) if via16 else src, nit, sb_layout, want_llr=True)
        assert np.array_equal(ref, out), "K=%d nit=%d snr=%g: %d code blocks differ" % (K, nit, snr, np.any(ref != out, axis=1).sum())
        assert np.array_equal(ref_llr, out_llr), "K=%d nit=%d snr=%g: decision LLRs differ" % (K, nit, snr)
    dec.free()


@pytest.mark.parametrize("K", [40, 416, 504, 816, 832, 1008, 1024, 2048, 2112, 3136, 6144])
def test_8bit_auto_all_regimes(hiplib, K):
    """AUTO through the 8-bit API: avx8 (32 sub-blocks), sse8 (16), or widened to sse16 / gen (turbodecoder.c:410-478).
    K=816 and 1008 have K%32 == 16: the last row of the exchanged vectors is subtracted wrapping, not saturating."""
    import srslte_amd as S
    from srslte_amd import capi

    for snr, scale in ((3.0, 12.0), (-1.0, 12.0), (0.0, 60.0), (-4.0, 8.0)):
        _check8(S, K, capi.TDEC_AUTO, O.ORC_TDEC_AUTO, 7, snr, (1, 2, 3, 4, 7, 8), seed=K * 5 + int(scale), scale=scale)


@pytest.mark.parametrize("K", [6144, 1024, 816])
def test_8bit_saturation_corner_cases(hiplib, K):
    """LLRs that are nothing but extremes (-128, -127, 127 and a few small values, no code word behind them): every saturating add of
    the 8-bit decoders hits a bound somewhere -- the device keeps int8 in the high byte of int16 halves and fixes the positive bound up
    after the fact (turbo_kernels.hip, Ar8); hard bits and decision LLRs must still equal the oracle's after every iteration count"""
    import srslte_amd as S
    from srslte_amd import capi

    rng = np.random.default_rng(K)
    n_cb = 6
    llr = rng.choice(np.array([-128, -127, 127, 127, -128, 0, 1, -1, 64, -64], np.int8), size=(n_cb, 3 * K + 12))
    llr[0] = 127
    llr[1] = -128
    dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO, llr8=True)
    for nit in (1, 2, 3, 8):
        ref, ref_llr = O.turbo_decode_8bit(llr, nit, K, O.ORC_TDEC_AUTO, 0, want_llr=True)
        out, out_llr = dec.decode(llr, nit, 0, want_llr=True)
        assert np.array_equal(ref, out), (K, nit)
        assert np.array_equal(ref_llr, out_llr), (K, nit)
    dec.free()


def test_8bit_every_block_size(hiplib):
    """all 188 block sizes through the 8-bit API (avx8 / sse8 windows, or widened to the 16-bit window / scalar decoders): noisy code
    words at an LLR scale that clips at +-127 most of the time, and inputs made of extremes only; hard bits and decision LLRs after
    1, 2, 3 and 8 half iterations equal the oracle's"""
    import srslte_amd as S
    from srslte_amd import capi

    rng = np.random.default_rng(1)
    bad = []
    for K in O.tc_sizes():
        n_cb = 3
        _, noisy = O.turbo_llrs_8bit(K, n_cb, -2.0, seed=K, scale=70.0)
        extreme = rng.choice(np.array([-128, -127, 127, 127, 0, 1, -1, 90, -90], np.int8), size=(n_cb, 3 * K + 12))
        dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO, llr8=True)
        for si, llr in enumerate((noisy, extreme)):
            for nit in (1, 2, 3, 8):
                ref, ref_llr = O.turbo_decode_8bit(llr, nit, K, O.ORC_TDEC_AUTO, 0, want_llr=True)
                out, out_llr = dec.decode(llr, nit, 0, want_llr=True)
                if not (np.array_equal(ref, out) and np.array_equal(ref_llr, out_llr)):
                    bad.append((K, si, nit))
        dec.free()
    assert not bad, bad[:10]


@pytest.mark.parametrize("impl,K", [("SSE8_WINDOW", 816), ("SSE8_WINDOW", 6144), ("AVX8_WINDOW", 1344), ("AVX8_WINDOW", 6144)])
def test_8bit_manual_implementations(hiplib, impl, K):
    import srslte_amd as S
    from srslte_amd import capi

    for via16 in (False, True):
        _check8(S, K, getattr(capi, "TDEC_" + impl), getattr(O, "ORC_TDEC_" + impl), 5, -1.0, (1, 2, 5, 8), seed=K, scale=30.0,
                via16=via16)
    h = C.c_void_p()
    assert S.lib().srsran_hip_tdec_batch_create(C.byref(h), 1280, 1, capi.TDEC_AVX8_WINDOW) == capi.SRSRAN_ERROR_INVALID_INPUTS


@pytest.mark.parametrize("K", [6144, 1024])
def test_8bit_rm_turbo_subblock_layout(hiplib, K):
    import srslte_amd as S
    from srslte_amd import capi

    _check8(S, K, capi.TDEC_AUTO, O.ORC_TDEC_AUTO, 5, -1.0, (1, 2, 8), seed=K + 1, sb_layout=1)


def test_8bit_handle_api(hiplib):
    """srsran_tdec_run_all_8bit / iteration_8bit through the drop-in handle"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    h = capi.Tdec()
    assert lib.srsran_tdec_init(C.byref(h), 6144) == 0
    lib.srsran_tdec_force_not_sb(C.byref(h))
    for K in (6144, 1024, 504, 40):
        msgs, llr = O.turbo_llrs_8bit(K, 2, -1.0, seed=K + 3)
        for nit in (1, 4, 8):
            ref = O.turbo_decode_8bit(llr, nit, K)
            for i in range(2):
                out = np.zeros(K // 8, np.uint8)
                assert lib.srsran_tdec_run_all_8bit(C.byref(h), O.P(llr[i].copy()), O.P(out), nit, K) == 0
                assert np.array_equal(out, ref[i])
        assert lib.srsran_tdec_new_cb(C.byref(h), K) == 0
        inp = llr[0].copy()
        for nit in range(1, 5):
            out = np.zeros(K // 8, np.uint8)
            lib.srsran_tdec_iteration_8bit(C.byref(h), O.P(inp), O.P(out))
            assert np.array_equal(out, O.turbo_decode_8bit(llr[:1], nit, K)[0]), (K, nit)
    lib.srsran_tdec_free(C.byref(h))
    # manual 8-bit decoder behind the 16-bit entry point, sub-block layout (what turbodecoder_test -d 7 does)
    h2 = capi.Tdec()
    assert lib.srsran_tdec_init_manual(C.byref(h2), 6144, capi.TDEC_AVX8_WINDOW) == 0
    K = 6144
    msgs, llr = O.turbo_llrs_8bit(K, 1, 0.0, seed=11)
    sb = O.natural_to_sb_layout(llr[0], K, 32).astype(np.int16)
    out = np.zeros(K // 8, np.uint8)
    assert lib.srsran_tdec_run_all(C.byref(h2), O.P(sb), O.P(out), 8, K) == 0
    assert np.array_equal(out, O.turbo_decode_8bit(llr, 8, K, O.ORC_TDEC_AVX8_WINDOW)[0])
    lib.srsran_tdec_free(C.byref(h2))


def test_saturating_llrs(hiplib):
    """LLRs near the int16 limits exercise the saturating (window) and wrapping (scalar) arithmetic"""
    import srslte_amd as S
    from srslte_amd import capi

    for K in (40, 512, 6144):
        _check_scaled(S, capi, K)


def _check_scaled(S, capi, K):
    msgs, llr = O.turbo_llrs(K, 6, 0.0, seed=K, scale=12000.0)
    dec = S.TdecBatch(K, 6, capi.TDEC_AUTO)
    ref, ref_llr = O.turbo_decode(llr, 8, K, want_llr=True)
    out, out_llr = dec.decode(llr, 8, want_llr=True)
    assert np.array_equal(ref, out) and np.array_equal(ref_llr, out_llr)


def test_ragged_batch_and_empty(hiplib):
    """batch sizes that do not fill a wave (8 code blocks per wave) and the zero-length corner"""
    import srslte_amd as S
    from srslte_amd import capi

    K = 6144
    msgs, llr = O.turbo_llrs(K, 17, 1.0, seed=5)
    ref = O.turbo_decode(llr, 8, K)
    dec = S.TdecBatch(K, 17, capi.TDEC_AUTO)
    for n in (1, 7, 8, 9, 17):
        out = dec.decode(llr[:n], 8)
        assert np.array_equal(ref[:n], out)
    rc = S.lib().srsran_hip_tdec_batch_run(dec._h, None, 0, None, 0, 0, 8, 0, None)
    assert rc == capi.SRSRAN_SUCCESS  # an empty batch is a no-op (a subframe without grants)
    h = C.c_void_p()
    assert S.lib().srsran_hip_tdec_batch_create(C.byref(h), 41, 1, capi.TDEC_AUTO) == capi.SRSRAN_ERROR_INVALID_INPUTS
    assert S.lib().srsran_hip_tdec_batch_create(C.byref(h), 40, 1, capi.TDEC_AVX_WINDOW) == capi.SRSRAN_ERROR_INVALID_INPUTS


def test_handle_api_matches_batch(hiplib):
    """srsran_tdec_init / run_all / iteration through the drop-in handle (host pointers)"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    h = capi.Tdec()
    assert lib.srsran_tdec_init(C.byref(h), 6144) == 0
    lib.srsran_tdec_force_not_sb(C.byref(h))
    for K in (6144, 1024, 40):
        msgs, llr = O.turbo_llrs(K, 2, -1.0, seed=K + 3)
        for nit in (1, 4, 8):
            ref = O.turbo_decode(llr, nit, K)
            for i in range(2):
                out = np.zeros(K // 8, np.uint8)
                inp = llr[i].copy()
                assert lib.srsran_tdec_run_all(C.byref(h), O.P(inp), O.P(out), nit, K) == 0
                assert np.array_equal(out, ref[i])
                assert lib.srsran_tdec_get_nof_iterations(C.byref(h)) == nit
        # iteration by iteration, as sch.c:420-454 drives it
        assert lib.srsran_tdec_new_cb(C.byref(h), K) == 0
        inp = llr[0].copy()
        for nit in range(1, 6):
            out = np.zeros(K // 8, np.uint8)
            lib.srsran_tdec_iteration(C.byref(h), O.P(inp), O.P(out))
            assert np.array_equal(out, O.turbo_decode(llr[:1], nit, K)[0]), (K, nit)
    assert lib.srsran_tdec_new_cb(C.byref(h), 7000) == -1
    lib.srsran_tdec_free(C.byref(h))
    assert h.max_long_cb == 0 and not h.dec16_hdlr[0]
    # sub-block layout expected by default in AUTO mode (no force_not_sb)
    h2 = capi.Tdec()
    assert lib.srsran_tdec_init(C.byref(h2), 6144) == 0
    K = 6144
    msgs, llr = O.turbo_llrs(K, 1, 0.0, seed=11)
    sb = O.natural_to_sb_layout(llr[0], K, 16)
    out = np.zeros(K // 8, np.uint8)
    assert lib.srsran_tdec_run_all(C.byref(h2), O.P(sb), O.P(out), 8, K) == 0
    assert np.array_equal(out, O.turbo_decode(llr, 8, K)[0])
    lib.srsran_tdec_free(C.byref(h2))


def test_full_size_roundtrip_property(hiplib):
    """BASELINE size (thousands of K=6144 blocks): encode -> noise-free LLR -> decode must return the message;
    every copy of the same block must decode identically (no cross-block interference in the batch)."""
    import srslte_amd as S
    from srslte_amd import capi

    K, n_cb = 6144, 4096
    rng = np.random.default_rng(9)
    base = rng.integers(0, 2, (8, K)).astype(np.uint8)
    # amplitude 40: the reference's saturating int16 metrics mis-decode a few bits of a NOISE-FREE block once
    # |LLR| is ~200 and 8 half iterations have run (the oracle shows the same errors); 40 is safely inside
    llr8 = np.stack([(40 * (2 * O.turbo_encode(b).astype(np.int32) - 1)).astype(np.int16) for b in base])
    llr = np.tile(llr8, (n_cb // 8, 1))
    out = S.TdecBatch(K, n_cb, capi.TDEC_AUTO).decode(llr, 8)
    bits = np.unpackbits(out, axis=1)
    assert np.array_equal(bits, np.tile(base, (n_cb // 8, 1)))
    # and the high-amplitude regime where the reference itself errs: identical errors, identical copies
    llr8 = np.stack([(200 * (2 * O.turbo_encode(b).astype(np.int32) - 1)).astype(np.int16) for b in base])
    out = S.TdecBatch(K, n_cb, capi.TDEC_AUTO).decode(np.tile(llr8, (n_cb // 8, 1)), 8)
    assert np.array_equal(out, np.tile(O.turbo_decode(llr8, 8, K), (n_cb // 8, 1)))
