#!/usr/bin/env python3
"""NR transport blocks received on the device: soft demodulation (int8) + sign change + descrambling (pdsch_nr.c:456-470) and then the
whole loop of sch_nr_decode (sch_nr.c:522-713) in one call -- segmentation, rate de-matching into the soft buffer, LDPC with CRC24B
early stop, code-block verdicts, payload assembly, transport CRC24A.  Transport blocks of 67,368 bits (8 code blocks, BG1, Z = 384,
256-QAM).  Prints one JSON line.  Single GPU; equalised symbols resident in HBM."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tbs", type=int, default=1024); ap.add_argument("--snr", type=float, default=24.0)
    ap.add_argument("--iters", type=int, default=10); ap.add_argument("--cpu-sample", type=int, default=1)
    a = ap.parse_args()
    import torch
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    lib = S.lib()
    dev = torch.device("cuda", 0)
    capi.check(lib.srsran_hip_set_device(0), "set_device")
    tbs, R, mod, Qm, Nl = 67368, 0.67, 4, 8, 1
    G = 8 * 12672
    cfg = O.sch_nr_tb_info(tbs, R, mod, G, Nl, 0)
    assert (cfg.bg, cfg.C, cfg.Z, cfg.F) == (0, 8, 384, 0)
    N = 66 * cfg.Z
    cfg.Nref = N
    ncb, nsym = cfg.C, G // Qm
    n_cb = a.tbs * ncb
    st = torch.cuda.current_stream().cuda_stream
    pool_tb = 8
    rng = np.random.default_rng(6)
    payload = rng.integers(0, 256, (pool_tb, tbs // 8)).astype(np.uint8)
    seeds = [int(rng.integers(0, 1 << 31)) for _ in range(pool_tb)]
    sigma = 10 ** (-a.snr / 20) / np.sqrt(2)
    syms = []
    for t in range(pool_tb):
        e = O.sch_nr_encode_tb(cfg, 0, payload[t])  # oracle restatement of sch_nr_encode
        x = O.modulate(e ^ O.sequence_bits(seeds[t], G), mod)
        syms.append((x + sigma * (rng.standard_normal(nsym) + 1j * rng.standard_normal(nsym))).astype(np.complex64))
    d_sym = torch.from_numpy(np.stack(syms).view(np.float32)).to(dev).repeat((a.tbs + pool_tb - 1) // pool_tb, 1)[:a.tbs].contiguous()
    d_llr = torch.zeros((a.tbs, G), dtype=torch.int8, device=dev)
    DS = 8448 // 8
    d_soft = torch.zeros((n_cb, N), dtype=torch.int8, device=dev)
    d_data = torch.zeros((n_cb, DS), dtype=torch.uint8, device=dev)
    d_pay = torch.zeros((a.tbs, tbs // 8), dtype=torch.uint8, device=dev)
    cb_crc = np.zeros(n_cb, np.uint8)
    dj = (capi.HipDemodJob * a.tbs)(*[capi.HipDemodJob(mod, nsym, i * nsym, i * G, seeds[i % pool_tb], 3) for i in range(a.tbs)])
    tb = (capi.HipNrTb * a.tbs)(*[capi.HipNrTb(R, tbs, mod, 0x100, Nl, G, 0, i * G, i * (tbs // 8), i * ncb, 0) for i in range(a.tbs)])
    res = (capi.HipNrTbResult * a.tbs)()
    hd, hn = C.c_void_p(), C.c_void_p()
    capi.check(lib.srsran_hip_demod_create(C.byref(hd)), "demod_create")
    capi.check(lib.srsran_hip_sch_nr_create(C.byref(hn), 0.8, a.iters, n_cb), "sch_nr_create")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t_demod = []

    def step():
        # first transmission: SRSRAN_HIP_NR_TB_NEW_DATA (0x100 in rv) stands for srsran_softbuffer_rx_reset
        cb_crc[:] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        capi.check(lib.srsran_hip_demod_run(hd, d_sym.data_ptr(), d_llr.data_ptr(), capi.LLR_BYTE, dj, a.tbs, st), "demod")
        ev[1].record()
        capi.check(lib.srsran_hip_sch_nr_decode(hn, d_llr.data_ptr(), tb, a.tbs, d_soft.data_ptr(), N, cb_crc.ctypes.data, d_data.data_ptr(), DS,
                                                d_pay.data_ptr(), res, st), "sch_nr_decode")
        dt = time.perf_counter() - t0
        t_demod.append(ev[0].elapsed_time(ev[1]))
        return dt

    for _ in range(a.warmup):
        step()
    t_demod.clear()
    dt = sum(step() for _ in range(a.steps)) / a.steps
    ok = sum(1 for r in res if r.crc_ok)
    got = d_pay[:pool_tb].cpu().numpy()
    good = all(np.array_equal(got[i], payload[i]) for i in range(pool_tb) if res[i].crc_ok)
    # CPU: the oracle's restatement of the same chain on the first transport block(s), single thread
    t1 = time.perf_counter()
    par = True
    for i in range(a.cpu_sample):
        llr = O.sequence_apply((-O.demod_soft(mod, syms[i], "b").astype(np.int32)).astype(np.int8), seeds[i])  # pdsch_nr.c:456-470
        soft, crc, data = np.zeros((ncb, N), np.int8), np.zeros(ncb, np.uint8), np.zeros((ncb, DS), np.uint8)
        out, okc, avg = O.sch_nr_decode_tb(cfg, 0, 0.8, a.iters, llr, soft, crc, data)
        par = (par and okc == res[i].crc_ok and abs(avg - res[i].avg_iter) < 1e-6 and np.array_equal(out, got[i])
               and np.array_equal(crc, cb_crc[i * ncb:(i + 1) * ncb]))
    tc = time.perf_counter() - t1
    out = {"metric": "NR transport blocks received, Mbit/s of TBS (256-QAM symbols -> int8 LLRs -> descrambling -> sch_nr_decode: de-matching, LDPC with CRC early stop, TB CRC)",
           "value": a.tbs * tbs / dt / 1e6, "unit": "Mbit/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3,
           "config": {"workload": "%d transport blocks of TBS %d (%d code blocks, BG1, Z=384, E=12672, 256-QAM), Es/N0 %.1f dB, max %d iterations"
                                  % (a.tbs, tbs, ncb, a.snr, a.iters)},
           "tb_crc_ok": [ok, a.tbs], "avg_iterations": float(np.mean([r.avg_iter for r in res])), "payload_matches_on_ok_blocks": bool(good),
           "demod_descramble_ms": float(np.mean(t_demod)),
           "cpu_baseline": {"value": a.cpu_sample * tbs / tc / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "port",
                            "sample": "%d transport block(s), oracle restatement of demodulate + descramble + sch_nr_decode (scalar C)" % a.cpu_sample},
           "parity_vs_oracle": "identical verdicts, iteration counts and payload" if par else "MISMATCH"}
    print(json.dumps(out))


main()
