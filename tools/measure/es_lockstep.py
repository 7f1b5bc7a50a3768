#!/usr/bin/env python3
"""What the lockstep of the 8 code blocks of a wave costs the early-stop decoder: per-block half-iteration counts (single-code-block transport
blocks through srsran_hip_sch_decode, so that avg_iterations IS the block's count), their histogram, the per-wave maximum the throughput
kernel pays for, and the same blocks ordered by their count (the bound of any re-grouping scheme)."""
import sys, os, ctypes as C, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
lib = S.lib(); lib.srsran_hip_set_device(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
tbs, Qm, G = 6120, 6, 9216  # one code block of K = 6144, rate 2/3
pool_n, n_tb, iters = 512, 16384, 10
out = {}
for snr in (6.0, 4.8):
    rng = np.random.default_rng(int(snr * 10))
    pool = [O.make_tb(tbs, Qm, G, 0, snr, rng)[0] for _ in range(pool_n)]
    e_pool = torch.from_numpy(np.stack(pool)).to(dev)
    dlen = tbs // 8 + 8
    h = C.c_void_p(); capi.check(lib.srsran_hip_sch_create(C.byref(h)), "create")
    def run(order, lat):
        lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", lat)
        d_e = e_pool[torch.from_numpy(order).to(dev)].contiguous()
        n = order.size
        d_data = torch.zeros((n, dlen), dtype=torch.uint8, device=dev)
        d_soft = torch.zeros((n, capi.SOFTBUFFER_CB_SIZE), dtype=torch.int16, device=dev)
        tb_arr = (capi.HipTb * n)(*[capi.HipTb(tbs, Qm, 0x100, G, i * G, i * dlen, i) for i in range(n)])
        res = (capi.HipTbResult * n)(); flags = np.zeros(n, np.uint8)
        best = 1e9
        for rep in range(3):
            flags[:] = 0; torch.cuda.synchronize(); t0 = time.perf_counter()
            capi.check(lib.srsran_hip_sch_decode(h, d_e.data_ptr(), tb_arr, n, iters, d_soft.data_ptr(), flags.ctypes.data, d_data.data_ptr(), res, st), "decode")
            best = min(best, time.perf_counter() - t0)
        return best, np.array([r.avg_iterations for r in res]), np.array([r.crc_ok for r in res])
    order = np.arange(n_tb) % pool_n
    t_nat, it, ok = run(order, b"0")
    it = np.round(it).astype(int)
    hist = np.bincount(it, minlength=iters + 1)
    wave_max = it.reshape(-1, 8).max(axis=1)
    order_sorted = order[np.argsort(it, kind="stable")]
    t_sorted, it2, _ = run(order_sorted, b"0")
    out["snr_%.1f" % snr] = {
        "blocks": n_tb, "distinct": pool_n, "blocks_ok": int((ok == 0).sum()), "mean_half_iterations": float(it.mean()),
        "histogram_half_iterations": {str(i): int(hist[i]) for i in range(1, iters + 1) if hist[i]},
        "mean_of_wave_maximum": float(wave_max.mean()), "lockstep_overhead": float(wave_max.mean() / it.mean() - 1.0),
        "ms_as_they_come": t_nat * 1e3, "ms_sorted_by_iteration_count": t_sorted * 1e3, "gain_of_perfect_grouping": float(t_nat / t_sorted - 1.0)}
    print(json.dumps(out["snr_%.1f" % snr]), flush=True)
    lib.srsran_hip_sch_free(h)
lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", None)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_es_lockstep.json"), "w"), indent=1)
