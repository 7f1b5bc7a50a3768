#!/usr/bin/env python3
"""time tdec_batch_run for different numbers of half iterations (phase cost breakdown)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
K, n_cb = 6144, int(sys.argv[1]) if len(sys.argv) > 1 else 53248
dev = torch.device("cuda", 0)
_, pool = O.turbo_llrs(K, 16, 0.0, seed=1)
d_llr = torch.from_numpy(pool).to(dev).repeat((n_cb + 15) // 16, 1)[:n_cb].contiguous()
d_bits = torch.zeros((n_cb, K // 8), dtype=torch.uint8, device=dev)
dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO)
st = torch.cuda.current_stream().cuda_stream
res = {}
for nit in (1, 2, 3, 4, 8, 16):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dec.run(d_llr, 3 * K + 12, d_bits, K // 8, n_cb, nit, 0, st); e1.record(); torch.cuda.synchronize()
        res[nit] = e0.elapsed_time(e1)
    print("nit=%2d  %.3f ms" % (nit, res[nit]), flush=True)
print("DEC1(n=0)+extract+decision: %.3f  DEC2: %.3f  DEC1(app): %.3f  per full iteration: %.3f" %
      (res[1], res[2] - res[1], res[3] - res[2], (res[16] - res[8]) / 4))
