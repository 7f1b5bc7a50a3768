"""Soft demodulation + descrambling (demod_soft.c, sequence.c:440-607) through the C ABI against the oracle and the
reference's recorded outputs: bit-exact int16 / int8 / float soft bits for the five modulations, including the places
where the x86 reference switches between its SIMD body and scalar tail rules."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FN = {"s": "srsran_demod_soft_demodulate_s", "b": "srsran_demod_soft_demodulate_b", "f": "srsran_demod_soft_demodulate"}


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def test_drop_in_vs_reference_fixture(hiplib):
    import srslte_amd as S

    lib = S.lib()
    d = np.load(os.path.join(G, "modem_ref.npz"))
    for key in d["cases"]:
        key = str(key)
        mod = int(key[1])
        x = np.ascontiguousarray(d[key + "_x"])
        for kind in "sbf":
            out = np.zeros(x.size * O.QM[mod], O.LLR_DTYPES[kind])
            assert getattr(lib, FN[kind])(mod, O.P(x), O.P(out), x.size) == 0
            assert _same(out, d[key + "_" + kind]), (key, kind)
    x = np.zeros(4, np.complex64)
    out = np.zeros(64, np.int16)
    assert lib.srsran_demod_soft_demodulate_s(5, O.P(x), O.P(out), 4) == -1  # demod_soft.c:888-891
    assert lib.srsran_demod_soft_demodulate_s(-1, O.P(x), O.P(out), 4) == -1


@pytest.mark.parametrize("mod", range(5))
def test_drop_in_vs_oracle_sizes(hiplib, mod):
    """lengths across tile (1024 symbols) and SIMD-group boundaries; amplitudes that saturate and wrap"""
    import srslte_amd as S

    lib = S.lib()
    for n in (1, 2, 3, 5, 9, 255, 256, 257, 1023, 1024, 1025, 4099, 20011):
        for scale in (1.0, 0.05, 60.0):
            x = O.qam_symbols(mod, n, seed=n + mod, snr_db=10.0, scale=scale)
            for kind in "sbf":
                out = np.zeros(n * O.QM[mod], O.LLR_DTYPES[kind])
                assert getattr(lib, FN[kind])(mod, O.P(x), O.P(out), n) == 0
                assert _same(out, O.demod_soft(mod, x, kind)), (mod, n, scale, kind)


def test_sequence_apply(hiplib):
    import srslte_amd as S

    lib = S.lib()
    rng = np.random.default_rng(3)
    d = np.load(os.path.join(G, "modem_ref.npz"))
    for seed, L in d["seqs"]:
        seed, L = int(seed), int(L)
        c = np.unpackbits(d["seq_%d_%d" % (seed, L)])[:L]
        one = np.ones(L, np.int16)
        out = np.zeros(L, np.int16)
        lib.srsran_sequence_apply_s(O.P(one), O.P(out), L, seed)
        assert np.array_equal(out == -1, c == 1), (seed, L)
    for L in (1, 31, 511, 512, 513, 8191, 8192, 8193, 100003, 1 << 21):
        seed = int(rng.integers(0, 1 << 31))
        for dt, fn, lo, hi in ((np.int16, lib.srsran_sequence_apply_s, -32768, 32768), (np.int8, lib.srsran_sequence_apply_c, -128, 128)):
            x = rng.integers(lo, hi, L).astype(dt)
            out = np.zeros_like(x)
            fn(O.P(x), O.P(out), L, seed)
            assert np.array_equal(out, O.sequence_apply(x, seed)), (L, dt)
        x = rng.normal(size=L).astype(np.float32)
        x[:3] = [0.0, -0.0, np.inf][:min(3, L)]
        out = np.zeros_like(x)
        lib.srsran_sequence_apply_f(O.P(x), O.P(out), L, seed)
        assert _same(out, O.sequence_apply(x, seed)), L
    # channel seeds (sequences.c:63-66,116-119) and in-place operation as pusch.c:436-443 uses it
    for rnti, nslot, cell, q in ((0x1234, 4, 301, 0), (0xFFFF, 19, 503, 1)):
        x = rng.integers(-3000, 3000, 5000).astype(np.int16)
        y = x.copy()
        lib.srsran_sequence_pusch_apply_s(O.P(y), O.P(y), rnti, nslot, cell, y.size)
        assert np.array_equal(y, O.sequence_apply(x, O.pusch_seed(rnti, nslot, cell)))
        y = x.copy()
        lib.srsran_sequence_pdsch_apply_s(O.P(y), O.P(y), rnti, q, nslot, cell, y.size)
        assert np.array_equal(y, O.sequence_apply(x, O.pdsch_seed(rnti, q, nslot, cell)))
        assert lib.srsran_hip_sequence_pusch_seed(rnti, nslot, cell) == O.pusch_seed(rnti, nslot, cell)
        assert lib.srsran_hip_sequence_pdsch_seed(rnti, q, nslot, cell) == O.pdsch_seed(rnti, q, nslot, cell)
        x8 = rng.integers(-128, 128, 777).astype(np.int8)
        y8 = x8.copy()
        lib.srsran_sequence_pusch_apply_c(O.P(y8), O.P(y8), rnti, nslot, cell, y8.size)
        assert np.array_equal(y8, O.sequence_apply(x8, O.pusch_seed(rnti, nslot, cell)))
    too_long = np.zeros((1 << 21) + 1, np.int8)
    out = np.ones_like(too_long)
    lib.srsran_sequence_apply_c(O.P(too_long), O.P(out), too_long.size, 1)  # refused loudly, output untouched
    assert out.all()


@pytest.mark.parametrize("kind", ["s", "b", "f"])
def test_fused_batch(hiplib, kind):
    """one launch over many jobs of mixed modulation / length / alignment; demodulate + descramble == the two reference steps"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(11)
    dt = O.LLR_DTYPES[kind]
    llr_type = {"s": capi.LLR_SHORT, "b": capi.LLR_BYTE, "f": capi.LLR_FLOAT}[kind]
    jobs, syms, want = [], [], []
    s_off = l_off = 0
    for i in range(40):
        mod = int(rng.integers(0, 5))
        n = int(rng.choice([1, 7, 144, 1200, 1024, 3000, 14400]))
        x = O.qam_symbols(mod, n, seed=100 + i, snr_db=15.0)
        seed = int(rng.integers(0, 1 << 31))
        scr = (1, 1, 2, 3, 0)[i % 5]  # bit 0 descramble, bit 1 sign change first (srsran_vec_neg_bb, pdsch_nr.c:467)
        llr = O.demod_soft(mod, x, kind)
        if scr & 2:
            llr = -llr if kind == "f" else (-llr.astype(np.int32)).astype(dt)
        if scr & 1:
            llr = O.sequence_apply(llr, seed)
        pad = int(rng.choice([0, 0, 1, 3, 8]))  # some jobs start unaligned
        jobs.append((mod, n, s_off, l_off + pad, seed, scr))
        syms.append(x)
        want.append((l_off + pad, llr))
        s_off += n
        l_off += pad + llr.size
        l_off += (-l_off) % 16 if i % 2 else 0
    all_sym = np.concatenate(syms)
    d_sym = S.DeviceBuffer.from_numpy(all_sym)
    d_llr = S.DeviceBuffer.from_numpy(np.zeros(l_off + 64, dt))
    arr = (capi.HipDemodJob * len(jobs))(*[capi.HipDemodJob(*j) for j in jobs])
    h = C.c_void_p()
    capi.check(lib.srsran_hip_demod_create(C.byref(h)), "create")
    for _ in range(2):  # second call re-uses the handle's job buffers
        capi.check(lib.srsran_hip_demod_run(h, d_sym.ptr, d_llr.ptr, llr_type, arr, len(jobs), None), "run")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    got = d_llr.to_numpy(dt, (l_off + 64,))
    covered = np.zeros(got.size, bool)
    for (off, llr), j in zip(want, jobs):
        assert _same(got[off:off + llr.size], llr), j
        covered[off:off + llr.size] = True
    assert not got[~covered].view(np.uint8 if kind != "f" else np.uint32).any()  # nothing written outside the jobs
    # descrambling-only jobs (SRSRAN_HIP_MOD_NONE) over existing soft bits, in place
    x = (rng.normal(size=50000) * 50).astype(dt)
    d = S.DeviceBuffer.from_numpy(x)
    j2 = (capi.HipDemodJob * 2)(capi.HipDemodJob(capi.MOD_NONE, 30000, 0, 0, 99, 1), capi.HipDemodJob(capi.MOD_NONE, 19999, 30001, 30001, 7, 1))
    capi.check(lib.srsran_hip_demod_run(h, d.ptr, d.ptr, llr_type, j2, 2, None), "run")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    got = d.to_numpy(dt, x.shape)
    assert _same(got[:30000], O.sequence_apply(x[:30000], 99)) and _same(got[30001:], O.sequence_apply(x[30001:], 7))
    assert _same(got[30000:30001], x[30000:30001])
    bad = (capi.HipDemodJob * 1)(capi.HipDemodJob(6, 10, 0, 0, 0, 0))
    assert lib.srsran_hip_demod_run(h, d.ptr, d.ptr, llr_type, bad, 1, None) == capi.SRSRAN_ERROR_INVALID_INPUTS
    lib.srsran_hip_demod_free(h)


def test_predecoding_single(hiplib):
    """single-antenna equaliser (mimo/precoding.c:357-392) vs the oracle (double arithmetic) and the recorded reference outputs"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    rng = np.random.default_rng(21)
    for n in (1, 2, 7, 32, 33, 1200, 14401):
        y = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64)
        h = (0.3 + 0.7 * (rng.normal(size=n) + 1j * rng.normal(size=n))).astype(np.complex64)
        for noise in (0.0, 0.05):
            for scaling in (1.0, 0.7):
                x = np.zeros_like(y)
                assert lib.srsran_predecoding_single(O.P(y), O.P(h), O.P(x), None, n, scaling, noise) == n
                want = O.predecoding_single(y, h, scaling, noise)
                assert np.abs(x - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (n, noise, scaling)
                csi = np.zeros(n, np.float32)
                assert lib.srsran_predecoding_single(O.P(y), O.P(h), O.P(x), O.P(csi), n, scaling, noise) == n
                wx, wc = O.predecoding_single(y, h, scaling, noise, want_csi=True)
                assert np.abs(x - wx).max() <= 1e-6 * max(1.0, np.abs(wx).max()) and np.abs(csi - wc).max() <= 1e-6 * wc.max()
    d = np.load(os.path.join(G, "modem_ref.npz"))
    y, h = d["eq_y"], d["eq_h"]
    for i, (scaling, noise) in enumerate(d["eq_par"]):
        x = np.zeros_like(y)
        lib.srsran_predecoding_single(O.P(y), O.P(h), O.P(x), None, y.size, float(scaling), float(noise))
        assert np.abs(x - d["eq_x"][i]).max() <= 1e-6 * np.abs(d["eq_x"][i]).max()  # the reference's own float result
    # batched: device pointers
    n = 100000
    y = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64)
    h = (0.5 + 0.5 * (rng.normal(size=n) + 1j * rng.normal(size=n))).astype(np.complex64)
    dy, dh, dx = S.DeviceBuffer.from_numpy(y), S.DeviceBuffer.from_numpy(h), S.DeviceBuffer(n * 8)
    capi.check(lib.srsran_hip_predecoding_single(dy.ptr, dh.ptr, dx.ptr, None, n, 1.0, 0.01, None), "eq")
    capi.check(lib.srsran_hip_stream_sync(None), "sync")
    want = O.predecoding_single(y, h, 1.0, 0.01)
    assert np.abs(dx.to_numpy(np.complex64, (n,)) - want).max() <= 1e-6 * np.abs(want).max()
