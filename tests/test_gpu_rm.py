"""Turbo rate de-matching (srsran_rm_turbo_rx_lut and friends, rm_turbo.c:390-478) through the C ABI against the oracle:
bit-exact accumulation into the soft buffer, for every receiver layout, wrap-around (repetition) and HARQ combining."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _orc_rx(x, base, K, rv, nsb):
    L = O.orc()
    fn = L.orc_rm_turbo_rx_8bit if x.dtype == np.int8 else L.orc_rm_turbo_rx
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    out = base.copy()
    assert fn(O.P(x), O.P(out), x.size, K, rv, nsb) == 0
    return out


@pytest.mark.parametrize("K", [40, 104, 512, 1008, 2112, 6144])
def test_batch_vs_oracle(hiplib, K):
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(K)
    n_cb = 5
    for dtype, fn in ((np.int16, lib.srsran_hip_rm_turbo_rx_batch), (np.int8, lib.srsran_hip_rm_turbo_rx_batch_8bit)):
        amp = 3000 if dtype == np.int16 else 60
        for nsb in (0, 8, 16, 32):
            if nsb and (K % nsb or K // nsb <= 40):
                continue
            n_out = 3 * (K + 32) + 12
            for rv in range(4):
                for E in (K // 2 + 5, 3 * K + 12, 3 * K + 100, 7 * K + 33):
                    x = rng.integers(-amp, amp, (n_cb, E)).astype(dtype)
                    base = rng.integers(-amp // 8, amp // 8, (n_cb, n_out)).astype(dtype)
                    d_in, d_out = S.DeviceBuffer.from_numpy(x), S.DeviceBuffer.from_numpy(base)
                    capi.check(fn(d_in.ptr, E, E, d_out.ptr, n_out, n_cb, K, rv, nsb, None), "rm batch")
                    capi.check(lib.srsran_hip_stream_sync(None), "sync")
                    got = d_out.to_numpy(dtype, (n_cb, n_out))
                    for i in range(n_cb):
                        assert np.array_equal(got[i], _orc_rx(x[i], base[i], K, rv, nsb)), (K, nsb, rv, E, i)
    d = S.DeviceBuffer(64)
    assert lib.srsran_hip_rm_turbo_rx_batch(d.ptr, 1, 1, d.ptr, 1, 1, 41, 0, 0, None) == capi.SRSRAN_ERROR_INVALID_INPUTS
    assert lib.srsran_hip_rm_turbo_rx_batch(d.ptr, 1, 1, d.ptr, 1, 1, 40, 4, 0, None) == capi.SRSRAN_ERROR_INVALID_INPUTS


def test_drop_in_and_harq_chain(hiplib):
    """srsran_rm_turbo_rx_lut on host pointers; two HARQ transmissions (rv 0 then rv 2) of a punctured code block combine
    in the soft buffer and the turbo decoder, fed with that buffer in its sub-block layout, recovers the message"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(5)
    sizes = O.tc_sizes()
    for K in (40, 1024, 6144):
        ci = sizes.index(K)
        msg = rng.integers(0, 2, K).astype(np.uint8)
        d = O.turbo_encode(msg)  # natural [d0 d1 d2] x K + 12 tail bits == the natural rate-matching buffer order
        nsb = lib.srsran_tdec_autoimp_get_subblocks(K)
        n_out = 3 * (K + 32) + 12
        soft = np.zeros(n_out, np.int16)
        ref = soft.copy()
        E = int(1.2 * K) // 2 * 2
        sigma = 0.9
        for rv in (0, 2):
            t = np.zeros(3 * K + 12, np.uint16)
            assert lib.srsran_hip_rm_turbo_table(O.P(t), K, rv, 0) == 0
            tx = d[t[np.arange(E) % t.size]]  # transmit side of 36.212 5.1.4.1.2: read the circular buffer from k0
            y = (2.0 * tx - 1.0) + sigma * rng.standard_normal(E)
            e = np.clip(np.round(40 * y), -32768, 32767).astype(np.int16)
            assert lib.srsran_rm_turbo_rx_lut(O.P(e), O.P(soft), E, ci, rv) == 0
            ref = _orc_rx(e, ref, K, rv, nsb)
            assert np.array_equal(soft, ref)
        h = capi.Tdec()
        assert lib.srsran_tdec_init(C.byref(h), 6144) == 0
        out = np.zeros(K // 8, np.uint8)
        assert lib.srsran_tdec_run_all(C.byref(h), O.P(soft.copy()), O.P(out), 8, K) == 0
        assert np.array_equal(np.unpackbits(out), msg), K
        lib.srsran_tdec_free(C.byref(h))
    x8 = rng.integers(-50, 50, 700).astype(np.int8)
    s8 = np.zeros(3 * (504 + 32) + 12, np.int8)
    assert lib.srsran_rm_turbo_rx_lut_8bit(O.P(x8), O.P(s8), 700, sizes.index(504), 1) == 0
    assert np.array_equal(s8, _orc_rx(x8, np.zeros_like(s8), 504, 1, lib.srsran_tdec_autoimp_get_subblocks_8bit(504)))
    assert lib.srsran_rm_turbo_rx_lut(O.P(x8), O.P(s8), 10, 188, 0) == capi.SRSRAN_ERROR_INVALID_INPUTS
    lib.srsran_rm_turbo_gentables()
    lib.srsran_rm_turbo_free_tables()
