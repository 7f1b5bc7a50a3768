"""The reference's threading contract under the unchanged handle API: one worker thread per in-flight subframe, each with its
OWN handles (srsenb/src/phy/lte/cc_worker.cc:212-231, lib/include/srsran/common/thread_pool.h:48), all calling at once.
Every result must equal the oracle's whether the calls were merged into batch launches by the submission queues
(srslte_amd/csrc/coalesce.h, default) or ran on the handles' private streams (srsran_hip_set_coalescing(0))."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_api as O

pytestmark = pytest.mark.gpu


def _stats(lib):
    b, u = C.c_uint64(), C.c_uint64()
    lib.srsran_hip_coalesce_stats(C.byref(b), C.byref(u))
    return b.value, u.value


def _run_threads(workers):
    errs = []

    def guard(f):
        try:
            f()
        except BaseException as e:  # noqa: BLE001 -- reported in the main thread
            errs.append(e)

    th = [threading.Thread(target=guard, args=(w,)) for w in workers]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]


@pytest.mark.parametrize("coalesce", [1, 0])
def test_concurrent_handles_all_results_equal_the_oracle(hiplib, coalesce):
    """15 threads at once: 3 threads in the handle-less rate de-matcher / soft demodulator / descrambler, 5 turbo decoders (K = 6144 x3 incl. one 8-bit, 1024, 40), 4 OFDM objects (100 PRB rx x2, 6 PRB rx, 100 PRB tx),
    3 LDPC decoders (BG1 Z=384 x2, one of them with CRC early stop; BG2 Z=96); every thread makes `calls` calls on its own handle"""
    import srslte_amd as S
    from srslte_amd import capi

    lib = hiplib
    lib.srsran_hip_set_coalescing(coalesce)
    b0, u0 = _stats(lib)
    calls = 6
    workers = []
    n_queued = [0]  # turbo calls: always through a queue
    n_maybe = [0]   # LDPC calls: through the queue only while more than four callers are inside the decoder

    def tdec_worker(K, llr8, seed, nit):
        if llr8:
            _, llr = O.turbo_llrs_8bit(K, calls, 1.0, seed)
            ref = O.turbo_decode_8bit(llr, nit, K)
        else:
            _, llr = O.turbo_llrs(K, calls, 0.5, seed)
            ref = O.turbo_decode(llr, nit, K)

        def run():
            h = capi.Tdec()
            assert lib.srsran_tdec_init(C.byref(h), 6144) == 0
            lib.srsran_tdec_force_not_sb(C.byref(h))  # natural-order input, as turbodecoder_test.c does
            out = np.zeros(K // 8, np.uint8)
            for i in range(calls):
                x = llr[i].copy()
                f = lib.srsran_tdec_run_all_8bit if llr8 else lib.srsran_tdec_run_all
                assert f(C.byref(h), O.P(x), O.P(out), nit, K) == 0
                assert np.array_equal(out, ref[i]), ("tdec", K, llr8, i)
                assert lib.srsran_tdec_get_nof_iterations(C.byref(h)) == nit
            lib.srsran_tdec_free(C.byref(h))

        n_queued[0] += calls
        return run

    def ofdm_worker(prb, N, tx, seed):
        ocfg = O.ofdm_cfg(prb, N, 0, 1)
        n, nsym, sf_sz, sf_re = O.ofdm_geometry(ocfg)
        rng = np.random.default_rng(seed)
        n_in, n_out = (sf_re, sf_sz) if tx else (sf_sz, sf_re)
        xs = ((rng.standard_normal((calls, n_in)) + 1j * rng.standard_normal((calls, n_in))) * 0.7).astype(np.complex64)
        ref = O.ofdm_tx(ocfg, xs) if tx else O.ofdm_rx(ocfg, xs)

        def run():
            bin_, bout = np.zeros(n_in, np.complex64), np.zeros(n_out, np.complex64)
            q = capi.Ofdm()
            cfg = capi.OfdmCfg()
            cfg.nof_prb, cfg.in_buffer, cfg.out_buffer, cfg.cp = prb, bin_.ctypes.data, bout.ctypes.data, capi.CP_NORM
            cfg.normalize, cfg.symbol_sz = True, N
            init = lib.srsran_ofdm_tx_init_cfg if tx else lib.srsran_ofdm_rx_init_cfg
            assert init(C.byref(q), C.byref(cfg)) == 0
            for i in range(calls):
                bin_[:] = xs[i]
                (lib.srsran_ofdm_tx_sf if tx else lib.srsran_ofdm_rx_sf)(C.byref(q))
                err = np.abs(bout - ref[i]).max() / max(1.0, float(np.sqrt(np.mean(np.abs(ref[i]) ** 2))))
                assert err < 1e-4, ("ofdm", prb, tx, i, err)
            (lib.srsran_ofdm_tx_free if tx else lib.srsran_ofdm_rx_free)(C.byref(q))

        return run  # (OFDM handles always use their private stream)

    def ldpc_worker(bg, Z, seed, with_crc):
        g = O.ldpc_graph(bg, Z)
        K, N = g.bgK * Z, g.bgN * Z
        poly, order = 0x1800063, 24
        rng = np.random.default_rng(seed)
        llrs, refs, rets = [], [], []
        for i in range(calls):
            msg = rng.integers(0, 2, K).astype(np.uint8)
            c = O.orc().orc_crc_bits(poly, order, O.P(msg), K - order)
            msg[K - order:] = [(c >> (order - 1 - j)) & 1 for j in range(order)]
            cw = np.zeros(N - 2 * Z, np.uint8)
            assert O.orc().orc_ldpc_encode(C.byref(g), O.P(msg), O.P(cw)) == 0
            sigma = 10 ** (-(2.5 if i % 2 else 0.0) / 20)
            llr = np.clip(np.round(((1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.size)) * 8 / sigma ** 2), -63, 63).astype(np.int8)
            r, rt = O.ldpc_decode(bg, Z, llr[None], 0.8, 9, None, crc=(poly, order) if with_crc else None)
            llrs.append(llr), refs.append(r[0]), rets.append(rt[0])

        def run():
            q = capi.LdpcDecoder()
            args = capi.LdpcDecoderArgs(capi.LDPC_C_AVX2, bg, Z, 0.8, 9)
            assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(args)) == 0
            crc = capi.Crc()
            crc.polynom, crc.order = poly, order
            out = np.zeros(K, np.uint8)
            for i in range(calls):
                if with_crc:
                    ret = lib.srsran_ldpc_decoder_decode_crc_c(C.byref(q), O.P(llrs[i]), O.P(out), N - 2 * Z, C.byref(crc))
                else:
                    ret = lib.srsran_ldpc_decoder_decode_c(C.byref(q), O.P(llrs[i]), O.P(out), N - 2 * Z)
                assert ret == rets[i], ("ldpc", bg, Z, i, ret, rets[i])
                if ret > 0:
                    assert np.array_equal(out, refs[i]), ("ldpc", bg, Z, i)
            lib.srsran_ldpc_decoder_free(C.byref(q))

        n_maybe[0] += calls
        return run

    def stateless_worker(seed):
        """the handle-less host-pointer functions (srsran_rm_turbo_rx_lut, srsran_demod_soft_demodulate_s, srsran_sequence_apply_s) keep a
        staging context per calling thread behind two process-wide table caches: three threads use them at once"""
        rng = np.random.default_rng(seed)
        K, rv = (6144, 0) if seed % 2 else (1024, 2)
        ci = O.tc_sizes().index(K)
        nsb = lib.srsran_tdec_autoimp_get_subblocks(K)
        E = 3 * K + 500 + seed
        es = [rng.integers(-200, 200, E).astype(np.int16) for _ in range(calls)]
        t = np.zeros(3 * K + 12, np.uint16)
        assert lib.srsran_hip_rm_turbo_table(O.P(t), K, rv, nsb) == 0
        L = 3000 + 7 * seed
        syms = [O.qam_symbols(3, L, seed + i, snr_db=15.0) for i in range(calls)]
        seq_seed = 12345 + seed

        def run():
            for i in range(calls):
                soft = np.zeros(3 * (K + 32) + 12, np.int16)
                assert lib.srsran_rm_turbo_rx_lut(O.P(es[i]), O.P(soft), E, ci, rv) == 0
                ref = np.zeros_like(soft)
                np.add.at(ref, t[np.arange(E) % t.size], es[i])  # out[T[i mod n_out]] += in[i] (rm_turbo.c:412-440); no wrap at these amplitudes
                assert np.array_equal(soft, ref), ("rm", seed, i)
                llr = np.zeros(6 * L, np.int16)
                assert lib.srsran_demod_soft_demodulate_s(3, O.P(syms[i]), O.P(llr), L) == 0
                assert np.array_equal(llr, O.demod_soft(3, syms[i], "s")), ("demod", seed, i)
                out = np.zeros_like(llr)
                lib.srsran_sequence_apply_s(O.P(llr), O.P(out), llr.size, seq_seed)
                assert np.array_equal(out, O.sequence_apply(llr, seq_seed)), ("seq", seed, i)

        return run

    workers += [tdec_worker(6144, False, 1, 8), tdec_worker(6144, False, 2, 8), tdec_worker(6144, True, 3, 8), tdec_worker(1024, False, 4, 5),
                tdec_worker(40, False, 5, 8)]
    workers += [stateless_worker(21), stateless_worker(22), stateless_worker(23)]
    workers += [ofdm_worker(100, 2048, False, 6), ofdm_worker(100, 2048, False, 7), ofdm_worker(6, 0, False, 8), ofdm_worker(100, 2048, True, 9)]
    workers += [ldpc_worker(0, 384, 10, False), ldpc_worker(0, 384, 11, True), ldpc_worker(1, 96, 12, False)]
    try:
        _run_threads(workers)
        b1, u1 = _stats(lib)
        if coalesce:
            assert n_queued[0] <= u1 - u0 <= n_queued[0] + n_maybe[0]  # every turbo call went through a queue
            assert 0 < b1 - b0 <= u1 - u0                              # ... in at most as many launches
        else:
            assert (b1, u1) == (b0, u0)            # private streams only
    finally:
        lib.srsran_hip_set_coalescing(1)


def test_same_shape_calls_are_merged_and_resumable(hiplib):
    """16 threads decoding K=6144 blocks at once share launches (fewer batches than calls); a run that went through the queue can be
    continued with srsran_tdec_iteration on the same handle (the private object re-runs the earlier half iterations first)"""
    from srslte_amd import capi

    lib = hiplib
    lib.srsran_hip_set_coalescing(1)
    K, nit, calls, n_thr = 6144, 8, 12, 16
    _, llr = O.turbo_llrs(K, n_thr, 0.0, 99)
    ref = O.turbo_decode(llr, nit, K)
    b0, u0 = _stats(lib)
    gate = threading.Barrier(n_thr)  # every round of calls is submitted at once: 16 requests for the queue's 4 lanes

    def worker(t):
        def run():
            h = capi.Tdec()
            assert lib.srsran_tdec_init(C.byref(h), K) == 0
            lib.srsran_tdec_force_not_sb(C.byref(h))
            out = np.zeros(K // 8, np.uint8)
            for _ in range(calls):
                x = llr[t].copy()
                gate.wait(timeout=120)
                assert lib.srsran_tdec_run_all(C.byref(h), O.P(x), O.P(out), nit, K) == 0
                assert np.array_equal(out, ref[t]), t
            lib.srsran_tdec_free(C.byref(h))
        return run

    _run_threads([worker(t) for t in range(n_thr)])
    b1, u1 = _stats(lib)
    assert u1 - u0 == n_thr * calls
    assert b1 - b0 < u1 - u0, "16 threads calling at once never shared a launch"
    # resume after a queued run
    h = capi.Tdec()
    assert lib.srsran_tdec_init(C.byref(h), K) == 0
    lib.srsran_tdec_force_not_sb(C.byref(h))
    out = np.zeros(K // 8, np.uint8)
    x = llr[3].copy()
    assert lib.srsran_tdec_run_all(C.byref(h), O.P(x), O.P(out), 3, K) == 0
    assert np.array_equal(out, O.turbo_decode(llr[3:4], 3, K)[0])
    lib.srsran_tdec_iteration(C.byref(h), O.P(x), O.P(out))
    assert lib.srsran_tdec_get_nof_iterations(C.byref(h)) == 4
    assert np.array_equal(out, O.turbo_decode(llr[3:4], 4, K)[0])
    lib.srsran_tdec_iteration(C.byref(h), O.P(x), O.P(out))
    assert np.array_equal(out, O.turbo_decode(llr[3:4], 5, K)[0])
    lib.srsran_tdec_free(C.byref(h))


def test_handle_initialised_on_one_thread_runs_on_another(hiplib):
    """handles are initialised on the main thread, srsran_hip_set_device is called, and the first run comes from a FRESH worker thread
    (no HIP call made on it yet): the queue's lanes must be created after the thread is bound to the process's device
    (turbo_host.cpp: tdec_run_all_queued; the LDPC path binds in ldpc_decode_any)"""
    from srslte_amd import capi

    lib = hiplib
    lib.srsran_hip_set_coalescing(1)
    K, nit = 2048, 6  # a shape no other test has queued before
    _, llr = O.turbo_llrs(K, 2, 1.0, 5)
    ref = O.turbo_decode(llr, nit, K)
    h = capi.Tdec()
    assert lib.srsran_tdec_init(C.byref(h), K) == 0
    lib.srsran_tdec_force_not_sb(C.byref(h))
    assert lib.srsran_hip_set_device(0) == 0
    out = np.zeros((2, K // 8), np.uint8)

    def run():
        for i in range(2):
            assert lib.srsran_tdec_run_all(C.byref(h), O.P(llr[i].copy()), O.P(out[i]), nit, K) == 0

    _run_threads([run])
    assert np.array_equal(out, ref)
    lib.srsran_tdec_free(C.byref(h))


def test_submission_queue_registry_is_bounded(hiplib):
    """a long-running process walks through many kernel shapes (block sizes, iteration budgets, LDPC rate-matched lengths): the registry
    keeps at most 12 queues (least recently used idle ones are released) and LDPC calls that differ only in rate-matched length / CRC
    share ONE queue (they are grouped per batch by a tag, not by the registry key); results stay equal to the oracle's throughout"""
    from srslte_amd import capi

    lib = hiplib
    lib.srsran_hip_set_coalescing(1)
    sizes = [512, 576, 640, 704, 768, 832, 896, 960, 1024, 1088, 1152, 1216, 1280, 1344, 1408, 1472]
    for rep in range(2):
        for K in sizes:
            _, llr = O.turbo_llrs(K, 1, 2.0, K)
            ref = O.turbo_decode(llr, 4, K)
            h = capi.Tdec()
            assert lib.srsran_tdec_init(C.byref(h), K) == 0
            lib.srsran_tdec_force_not_sb(C.byref(h))
            out = np.zeros(K // 8, np.uint8)
            assert lib.srsran_tdec_run_all(C.byref(h), O.P(llr[0].copy()), O.P(out), 4, K) == 0
            assert np.array_equal(out, ref[0]), K
            lib.srsran_tdec_free(C.byref(h))
            assert lib.srsran_hip_coalesce_shapes() <= 12
    # LDPC: 8 threads, four rate-matched lengths at once -> one queue for the (type, bg, Z, sf, iterations) engine
    bg, Z, nit, n_thr, calls = 1, 96, 6, 8, 5
    g = O.ldpc_graph(bg, Z)
    N = g.bgN * Z
    lens = [N - 2 * Z, (g.bgK + 9) * Z + 3, (g.bgK + 20) * Z, (g.bgK + 5) * Z]
    _, llrs = O.ldpc_llrs(bg, Z, n_thr, 2.0, seed=11)
    shapes0 = lib.srsran_hip_coalesce_shapes()
    b0, u0 = _stats(lib)

    def worker(t):
        def run():
            rm = lens[t % len(lens)]
            ref, _ = O.ldpc_decode(bg, Z, llrs[t:t + 1], 0.8, nit, rm)
            q = capi.LdpcDecoder()
            a = capi.LdpcDecoderArgs(capi.LDPC_C_AVX2, bg, Z, 0.8, nit)
            assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(a)) == 0
            out = np.zeros(g.bgK * Z, np.uint8)
            for _ in range(calls):
                assert lib.srsran_ldpc_decoder_decode_c(C.byref(q), O.P(llrs[t]), O.P(out), rm) == nit
                assert np.array_equal(out, ref[0]), (t, rm)
            lib.srsran_ldpc_decoder_free(C.byref(q))
        return run

    _run_threads([worker(t) for t in range(n_thr)])
    assert lib.srsran_hip_coalesce_shapes() <= max(shapes0 + 1, 12)
    assert lib.srsran_hip_coalesce_shapes() <= 12


def test_transport_block_seam_under_concurrent_workers(hiplib, tmp_path):
    """the reference's threading model on the seam, from C (tools/probe/seam_threads.c: 1 / 2 / 3 / 4 / 8 PHY worker threads, each with its own soft buffer,
    decoding a transport block per call through srsran_hip_decode_tb_cb at the same time; the block comes from the library's own transmit entry
    srsran_hip_encode_tb): every call of every thread returns the payload"""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "seam_threads")
    libdir = os.path.join(root, "srslte_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "probe", "seam_threads.c"), "-o", exe, "-L" + libdir,
                           "-lsrsran_phy_hip", "-Wl,-rpath," + libdir, "-lpthread", "-lm"])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    rows = re.findall(r"(\d+) worker thread\(s\).*?(\d+) of (\d+) decoded", out.stdout)
    assert [int(r[0]) for r in rows] == [1, 2, 3, 4, 8], out.stdout[-2000:]
    assert all(a == b for _, a, b in rows), rows
