#!/usr/bin/env python3
"""time tdec batch (natural-order int16 input, 8 half iterations) for several K"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
n_cb = 53248
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for K in [int(a) for a in sys.argv[1:]] or [6144, 5824, 5696, 2048, 1024, 1008]:
    _, pool = O.turbo_llrs(K, 16, 0.0, seed=1)
    d_llr = torch.from_numpy(pool).to(dev).repeat((n_cb + 15) // 16, 1)[:n_cb].contiguous()
    d_bits = torch.zeros((n_cb, (K + 7) // 8), dtype=torch.uint8, device=dev)
    dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO)
    for nit in (1, 8):
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dec.run(d_llr, 3 * K + 12, d_bits, (K + 7) // 8, n_cb, nit, 0, st); e1.record(); torch.cuda.synchronize()
        print("K=%5d nit=%d  %.3f ms  (%.2f ns per block and half iteration)" % (K, nit, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e6 / n_cb / nit), flush=True)
    del dec
