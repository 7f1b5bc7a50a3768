#!/usr/bin/env python3
"""BASELINE config 3 (5G NR 100 MHz): OFDM rx N=4096 / 273 PRB + LDPC BG1 Z=384 int8 layered min-sum, 20 iterations.
Prints one JSON line (same fields as bench.py).  Single GPU; device-resident inputs."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cw", type=int, default=16384); ap.add_argument("--slots", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20); ap.add_argument("--cpu-sample", type=int, default=24)
    a = ap.parse_args()
    import torch, ctypes as C
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    dev = torch.device("cuda", 0)
    S.capi.check(S.lib().srsran_hip_set_device(0), "set_device")
    Z, bg = 384, 0
    msgs, pool = O.ldpc_llrs(bg, Z, 32, 3.0, seed=11)
    n_llr = pool.shape[1]
    d_llr = torch.from_numpy(pool).to(dev).repeat((a.cw + 31) // 32, 1)[:a.cw].contiguous()
    d_msg = torch.zeros((a.cw, 22 * Z), dtype=torch.uint8, device=dev)
    dec = S.LdpcBatch(bg, Z, 0.8, a.iters, a.cw)
    ofdm = S.OfdmBatch(273, False, 4096, normalize=True, keep_dc=True)
    d_time = torch.view_as_complex(torch.randn((a.slots, ofdm.sf_sz, 2), device=dev) * 0.7071)
    d_re = torch.zeros((a.slots, ofdm.sf_re), dtype=torch.complex64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def step(ev=None):
        if ev: ev[0].record()
        ofdm.run(d_time, d_re, a.slots, st)
        if ev: ev[1].record()
        dec.run(d_llr, n_llr, d_msg, 22 * Z, a.cw, n_llr, None, st)
        if ev: ev[2].record()
    for _ in range(a.warmup): step()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps): step(evs[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_o = sum(e[0].elapsed_time(e[1]) for e in evs) / a.steps * 1e-3
    t_l = sum(e[1].elapsed_time(e[2]) for e in evs) / a.steps * 1e-3
    out = d_msg[:32].cpu().numpy()
    ref, _ = O.ldpc_decode(bg, Z, pool[:a.cpu_sample], 0.8, a.iters)
    t1 = time.perf_counter(); O.ldpc_decode(bg, Z, pool[:a.cpu_sample], 0.8, a.iters); tc = time.perf_counter() - t1
    cw_bytes = n_llr + 22 * Z
    sf_bytes = 8 * (15 * 4096 + 14 * 12 * 273)
    res = {"metric": "LDPC decoded Mbit/s (info bits, NR BG1 Z=384, 20 iterations) incl. OFDM demod N=4096", "value": a.cw * 22 * Z * a.steps / dt / 1e6,
           "unit": "Mbit/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
           "config": {"workload": "NR 100 MHz: ofdm_rx N=4096 273 PRB x %d slots(1 ms) + ldpc BG1 Z=384 %d it x %d code words" % (a.slots, a.iters, a.cw)},
           "ldpc_kernel_mbit_per_s": a.cw * 22 * Z / t_l / 1e6, "ldpc_ms": t_l * 1e3,
           "ofdm_msamples_per_s": a.slots * ofdm.sf_sz / t_o / 1e6, "ofdm_ms": t_o * 1e3,
           "roofline_ldpc": {"bound": "hbm", "achieved": a.cw * cw_bytes / t_l / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": a.cw * cw_bytes / t_l / 1e9 / 8000.0},
           "roofline_ofdm": {"bound": "hbm", "achieved": a.slots * sf_bytes / t_o / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": a.slots * sf_bytes / t_o / 1e9 / 8000.0},
           "cpu_baseline": {"value": a.cpu_sample * 22 * Z / tc / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "port", "sample": "%d code words, scalar C restatement" % a.cpu_sample},
           "parity_vs_oracle": "bit-exact" if np.array_equal(out[:a.cpu_sample], ref) else "MISMATCH"}
    print(json.dumps(res))
main()
