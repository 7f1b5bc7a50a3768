#!/usr/bin/env python3
"""latency kernel against the throughput kernel: ms per launch over batch sizes and half-iteration counts (fixed iterations, device-resident input);
slope = time per half iteration, intercept = input extraction + decision.  usage: lat_time.py [K] [sb_layout]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
sb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
llr8 = len(sys.argv) > 3 and sys.argv[3] == "8"
lib = S.lib(); lib.srsran_hip_set_device(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
_, pool = (O.turbo_llrs_8bit if llr8 else O.turbo_llrs)(K, 16, 1.0, seed=1)
if sb:
    pool = np.stack([O.natural_to_sb_layout(pool[i], K, 32 if llr8 else 16) for i in range(16)])
stride = pool.shape[1]
for n_cb in (1, 13, 104, 256, 512, 832, 1024, 2048, 4096, 8192):
    d_llr = torch.from_numpy(pool).to(dev).repeat((n_cb + 15) // 16, 1)[:n_cb].contiguous()
    d_bits = torch.zeros((n_cb, K // 8), dtype=torch.uint8, device=dev)
    dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO, llr8=llr8)
    row = []
    for lat, lat2 in ((b"0", b"0"), (b"1", b"0"), (b"1", b"1")):
        lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", lat)
        lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT2", lat2)
        for nit in (1, 2, 4, 8):
            best = 1e9
            for rep in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if llr8:
                    capi.check(lib.srsran_hip_tdec_batch_run_8bit(dec._h, d_llr.data_ptr(), stride, d_bits.data_ptr(), K // 8, n_cb, nit, sb, st), 'run8')
                else:
                    dec.run(d_llr, stride, d_bits, K // 8, n_cb, nit, sb, st)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            row.append(best)
    print("K=%d sb=%d n_cb=%5d  throughput kernel: %s   latency kernel: %s   two waves per block: %s  (ms at 1/2/4/8 half iterations)" %
          (K, sb, n_cb, " ".join("%.3f" % v for v in row[:4]), " ".join("%.3f" % v for v in row[4:8]), " ".join("%.3f" % v for v in row[8:])), flush=True)
    del dec
lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", None); lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT2", None)
