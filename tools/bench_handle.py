#!/usr/bin/env python3
"""The drop-in handle API as srsenb / srsue PHY workers call it (host pointers, one call = one code block or one subframe):
us per call and aggregate throughput of srsran_tdec_run_all (K=6144, 8 half iterations), srsran_ofdm_rx_sf (100 PRB, N=2048) and
srsran_ldpc_decoder_decode_c (BG1 Z=384, 20 iterations) from 1, 3 and 16 host threads, each thread with its OWN handle
(the reference's contract: one worker per in-flight subframe, cc_worker.cc:212-231) -- with the per-process submission queues
(csrc/coalesce.h) and with every handle on its private stream.  PCIe transfers are inside every call.  Prints one JSON line."""
import argparse, ctypes as C, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--threads", default="1,3,16")
    a = ap.parse_args()
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    lib = S.lib()
    capi.check(lib.srsran_hip_set_device(0), "set_device")
    K, nit = 6144, 8
    _, llr = O.turbo_llrs(K, 16, 1.0, seed=5)
    tref = O.turbo_decode(llr[:2], nit, K)
    bg, Z, lit = 0, 384, 20
    _, lllr = O.ldpc_llrs(bg, Z, 16, 2.0, seed=6)
    lref, _ = O.ldpc_decode(bg, Z, lllr[:2], 0.8, lit)
    ocfg = O.ofdm_cfg(100, 2048, 0, 1)
    n, nsym, sf_sz, sf_re = O.ofdm_geometry(ocfg)
    rng = np.random.default_rng(1)
    ox = ((rng.standard_normal((16, sf_sz)) + 1j * rng.standard_normal((16, sf_sz))) * 0.7).astype(np.complex64)
    oref = O.ofdm_rx(ocfg, ox[:2])

    def tdec_thread(t, calls, ok):
        h = capi.Tdec()
        assert lib.srsran_tdec_init(C.byref(h), K) == 0
        lib.srsran_tdec_force_not_sb(C.byref(h))
        x, out = llr[t % 16].copy(), np.zeros(K // 8, np.uint8)

        def run():
            for _ in range(calls):
                lib.srsran_tdec_run_all(C.byref(h), O.P(x), O.P(out), nit, K)
            ok[t] = t % 16 >= 2 or np.array_equal(out, tref[t % 16])
        return run, lambda: lib.srsran_tdec_free(C.byref(h))

    def ofdm_thread(t, calls, ok):
        bin_, bout = np.zeros(sf_sz, np.complex64), np.zeros(sf_re, np.complex64)
        q, cfg = capi.Ofdm(), capi.OfdmCfg()
        cfg.nof_prb, cfg.in_buffer, cfg.out_buffer, cfg.cp, cfg.normalize, cfg.symbol_sz = 100, bin_.ctypes.data, bout.ctypes.data, capi.CP_NORM, True, 2048
        assert lib.srsran_ofdm_rx_init_cfg(C.byref(q), C.byref(cfg)) == 0
        bin_[:] = ox[t % 16]

        def run():
            for _ in range(calls):
                lib.srsran_ofdm_rx_sf(C.byref(q))
            ok[t] = t % 16 >= 2 or np.abs(bout - oref[t % 16]).max() < 1e-4 * max(1.0, float(np.abs(oref[t % 16]).max()))
        return run, lambda: (lib.srsran_ofdm_rx_free(C.byref(q)), bin_, bout)

    def ldpc_thread(t, calls, ok):
        q = capi.LdpcDecoder()
        args = capi.LdpcDecoderArgs(capi.LDPC_C_AVX2, bg, Z, 0.8, lit)
        assert lib.srsran_ldpc_decoder_init(C.byref(q), C.byref(args)) == 0
        x, out = lllr[t % 16].copy(), np.zeros(22 * Z, np.uint8)

        def run():
            for _ in range(calls):
                lib.srsran_ldpc_decoder_decode_c(C.byref(q), O.P(x), O.P(out), 66 * Z)
            ok[t] = t % 16 >= 2 or np.array_equal(out, lref[t % 16])
        return run, lambda: lib.srsran_ldpc_decoder_free(C.byref(q))

    kinds = {"tdec_run_all K=6144 nit=8": (tdec_thread, K, "Mbit/s", a.calls),
             "ofdm_rx_sf 100 PRB N=2048": (ofdm_thread, sf_sz, "Msamples/s", a.calls),
             "ldpc_decoder_decode_c BG1 Z=384 20 it": (ldpc_thread, 22 * Z, "Mbit/s", max(20, a.calls // 4))}
    res = {"what": "handle API, host pointers, one unit per call, own handle per thread; H2D + kernel + D2H + synchronisation inside every call", "results": []}
    for name, (mk, unit_n, unit, calls) in kinds.items():
        for co in (1, 0):
            lib.srsran_hip_set_coalescing(co)
            for nt in [int(v) for v in a.threads.split(",")]:
                ok = [False] * nt
                pairs = [mk(t, calls, ok) for t in range(nt)]
                for run, _ in pairs[:1]:
                    pass
                # warm-up: one call per thread (creates the queue / private objects)
                warm = [mk(t, 1, [False] * nt) for t in range(nt)]
                for r, fr in warm:
                    r(); fr()
                b0, u0 = C.c_uint64(), C.c_uint64()
                lib.srsran_hip_coalesce_stats(C.byref(b0), C.byref(u0))
                th = [threading.Thread(target=r) for r, _ in pairs]
                t0 = time.perf_counter()
                for t in th: t.start()
                for t in th: t.join()
                dt = time.perf_counter() - t0
                b1, u1 = C.c_uint64(), C.c_uint64()
                lib.srsran_hip_coalesce_stats(C.byref(b1), C.byref(u1))
                for _, fr in pairs: fr()
                res["results"].append({"call": name, "coalescing": bool(co), "threads": nt, "calls_per_thread": calls,
                                       "us_per_call": dt / calls * 1e6, "aggregate": nt * calls * unit_n / dt / 1e6, "unit": unit,
                                       "units_per_launch": (u1.value - u0.value) / max(1, b1.value - b0.value) if co else 1.0,
                                       "results_ok": bool(all(ok))})
    lib.srsran_hip_set_coalescing(1)
    print(json.dumps(res))


main()
