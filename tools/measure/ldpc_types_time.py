#!/usr/bin/env python3
"""LDPC BG1 Z=384, 16,384 words, 20 iterations: the three LLR types (int8 packed kernel, int16 and float one-position kernels), all handing their
code words out through the work counter"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
S.lib().srsran_hip_set_device(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
bg, Z, cw, it = 0, 384, 16384, 20
g = O.ldpc_graph(bg, Z); K, N = g.bgK * Z, g.bgN * Z - 2 * Z
_, llrs = O.ldpc_llrs(bg, Z, 64, 3.0, seed=3)
for name, typ, dt, tdt in (("int8 (C)", capi.LDPC_C, np.int8, torch.int8), ("int16 (S)", capi.LDPC_S, np.int16, torch.int16), ("float (F)", capi.LDPC_F, np.float32, torch.float32)):
    d_llr = torch.from_numpy(llrs.astype(dt)).to(dev).repeat(cw // 64, 1).contiguous()
    d_msg = torch.zeros((cw, K), dtype=torch.uint8, device=dev)
    dec = S.LdpcBatch(bg, Z, 0.8, it, cw, dec_type=typ)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dec.run(d_llr, N, d_msg, K, cw, N, None, st); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-10s %.2f ms per %d words = %.2f Gbit/s of information bits" % (name, best, cw, cw * K / best / 1e6), flush=True)
    del dec
