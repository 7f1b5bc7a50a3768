#!/usr/bin/env python3
"""scalar decoder (K <= 400): one-lane-per-block kernel against the latency kernel (turbo_gen_lat_kernels.hip), ms per launch over batch sizes and half-iteration
counts (fixed iterations, device-resident input), outputs compared.  usage: gen_time.py [K ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
lib = S.lib(); lib.srsran_hip_set_device(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
for K in ([int(a) for a in sys.argv[1:]] or [40, 176, 400]):
    _, pool = O.turbo_llrs(K, 16, 1.0, seed=1)
    stride = pool.shape[1]
    for n_cb in (1, 8, 64, 512, 2048, 8192, 32768):
        d_llr = torch.from_numpy(pool).to(dev).repeat((n_cb + 15) // 16, 1)[:n_cb].contiguous()
        dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO)
        row, outs = [], []
        for lat in (b"0", b"1"):
            lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", lat)
            d_bits = torch.zeros((n_cb, K // 8), dtype=torch.uint8, device=dev)
            for nit in (1, 2, 4, 8):
                best = 1e9
                for rep in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    dec.run(d_llr, stride, d_bits, K // 8, n_cb, nit, 0, st)
                    e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1))
                row.append(best)
            outs.append(d_bits.cpu().numpy())
        print("K=%d n_cb=%5d  one lane per block: %s   latency kernel: %s  (ms at 1/2/4/8 half iterations)  same bytes: %s" %
              (K, n_cb, " ".join("%.3f" % v for v in row[:4]), " ".join("%.3f" % v for v in row[4:]), bool(np.array_equal(outs[0], outs[1]))), flush=True)
        del dec
lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", None)
