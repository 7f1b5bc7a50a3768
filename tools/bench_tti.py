#!/usr/bin/env python3
"""TTI-sized operating points of the transport-block decoder: what ONE worker call costs when it carries the transport blocks of one
subframe (the reference decodes one subframe per worker call: srsenb/src/phy/lte/cc_worker.cc:212-231, sch.c:389-492), not tens of
thousands of code blocks.

Per point (n_tb = 1 / 8 / 64 transport blocks of TBS 75,376 = 13 code blocks of 5,824 bits, 64-QAM, first transmission, CRC early stop, max
10 half iterations -- the reference's default) the call is timed END TO END FROM HOST MEMORY: pinned e bits -> H2D -> srsran_hip_sch_decode{,_8bit}
-> payload bytes D2H -> results on the host; p50 / p99 over --calls calls.  Two SNR knobs: a comfortable one (2 half iterations) and the
waterfall.  Beside it: the reference's own chain (oracle/_ref: srsran_rm_turbo_rx_lut + srsran_tdec_iteration + srsran_crc_checksum_byte in
the order of decode_tb_cb) on ONE host core for the same transport blocks, per transport block.  Writes one JSON document."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(round(p / 100.0 * (len(v) - 1))))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--snrs", default="8.0,6.0,4.8")
    ap.add_argument("--ntb", default="1,8,64")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default="")
    ap.add_argument("--llr8-mean", type=float, default=24.0, help="mean |LLR| of the int8 e bits (clipped to +-127)")
    a = ap.parse_args()
    import torch
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    lib = S.lib()
    dev = torch.device("cuda", 0)
    capi.check(lib.srsran_hip_set_device(0), "set_device")
    tbs, Qm, G = 75376, 6, 100800
    seg = O.cbsegm(tbs)
    ncb, dlen = seg["C"], tbs // 8 + 8
    st = torch.cuda.current_stream().cuda_stream
    doc = {"workload": "TBS %d (%d code blocks of %d), 64-QAM, G = %d e bits, first transmission, CRC early stop, max %d half iterations; "
                       "host e bits (pinned) -> H2D -> srsran_hip_sch_decode -> payload D2H, per call" % (tbs, ncb, seg["K1"], G, a.iters),
           "points": []}
    h = C.c_void_p()
    capi.check(lib.srsran_hip_sch_create(C.byref(h)), "sch_create")
    for snr in [float(x) for x in a.snrs.split(",")]:
        rng = np.random.default_rng(int(snr * 10))
        pool_n = 16
        pool = [O.make_tb(tbs, Qm, G, 0, snr, rng) for _ in range(pool_n)]
        for llr8 in (False, True):
            sdt, tdt = (np.int8, torch.int8) if llr8 else (np.int16, torch.int16)
            es = [np.clip(np.round(e * (a.llr8_mean / np.mean(np.abs(e)))), -127, 127).astype(np.int8) if llr8 else e for e, _ in pool]
            # the reference's chain on one core: per transport block
            ref_ms, ref_ok, ref_it = None, None, None
            if O.have_ref():
                chain = O.RefSchChain(llr8, a.iters)
                t_ref, oks, its = [], 0, []
                for i in range(pool_n):
                    soft, crc = chain.new_softbuffer(ncb), np.zeros(ncb, np.uint8)
                    soft[:] = 0
                    t0 = time.perf_counter()
                    ok, data, avg = chain.decode_tb(tbs, Qm, 0, es[i], soft, crc)
                    t_ref.append(time.perf_counter() - t0)
                    oks += ok
                    its.append(avg)
                ref_ms, ref_ok, ref_it = 1e3 * float(np.median(t_ref)), oks, float(np.mean(its))
            for n_tb in [int(x) for x in a.ntb.split(",")]:
                h_e = torch.from_numpy(np.stack([es[i % pool_n] for i in range(n_tb)])).pin_memory()
                d_e = torch.zeros((n_tb, G), dtype=tdt, device=dev)
                d_data = torch.zeros((n_tb, dlen), dtype=torch.uint8, device=dev)
                h_data = torch.zeros((n_tb, dlen), dtype=torch.uint8).pin_memory()
                d_soft = torch.zeros((n_tb * ncb, capi.SOFTBUFFER_CB_SIZE), dtype=tdt, device=dev)
                tb_arr = (capi.HipTb * n_tb)(*[capi.HipTb(tbs, Qm, 0x100, G, i * G, i * dlen, i * ncb) for i in range(n_tb)])
                res = (capi.HipTbResult * n_tb)()
                flags = np.zeros(n_tb * ncb, np.uint8)
                fn = lib.srsran_hip_sch_decode_8bit if llr8 else lib.srsran_hip_sch_decode

                def call():
                    flags[:] = 0
                    t0 = time.perf_counter()
                    d_e.copy_(h_e, non_blocking=True)
                    capi.check(fn(h, d_e.data_ptr(), tb_arr, n_tb, a.iters, d_soft.data_ptr(), flags.ctypes.data, d_data.data_ptr(), res, st), "sch_decode")
                    h_data.copy_(d_data, non_blocking=True)
                    torch.cuda.synchronize()
                    return time.perf_counter() - t0

                for _ in range(10):
                    call()
                ts = [call() for _ in range(a.calls)]
                ok = sum(1 for r in res if r.crc_ok == 0)
                avg_it = float(np.mean([r.avg_iterations for r in res]))
                good = all(np.array_equal(h_data[i].numpy()[:tbs // 8 + 3], pool[i % pool_n][1]) for i in range(n_tb) if res[i].crc_ok == 0)
                p = {"snr_knob_db": snr, "llr": "int8" if llr8 else "int16", "n_tb": n_tb, "code_blocks": n_tb * ncb,
                     "p50_ms": 1e3 * pct(ts, 50), "p99_ms": 1e3 * pct(ts, 99), "min_ms": 1e3 * min(ts),
                     "mbit_per_s_at_p50": n_tb * tbs / pct(ts, 50) / 1e6, "tb_ok": [ok, n_tb], "avg_half_iterations": avg_it,
                     "payload_matches": bool(good),
                     "reference_one_core": None if ref_ms is None else {"ms_per_tb_median": ref_ms, "tb_ok": [ref_ok, pool_n], "avg_half_iterations": ref_it,
                                                                        "ms_for_this_call_on_one_core": ref_ms * n_tb,
                                                                        "what": "oracle/_ref rm_turbo_rx_lut + tdec_iteration (AUTO) + crc, order of sch.c:389-466"}}
                doc["points"].append(p)
                sys.stderr.write(json.dumps(p) + "\n")
    txt = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, "w").write(txt)
    print(txt)


main()
