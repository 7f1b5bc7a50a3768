#!/usr/bin/env python3
"""one configuration of the latency kernel for counter passes: lat_one.py K sb n_cb nit lat(0/1)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
K, sb, n_cb, nit, lat = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5].encode()
lib = S.lib(); lib.srsran_hip_set_device(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
_, pool = O.turbo_llrs(K, 16, 1.0, seed=1)
if sb:
    pool = np.stack([O.natural_to_sb_layout(pool[i], K, 16) for i in range(16)])
d_llr = torch.from_numpy(pool).to(dev).repeat((n_cb + 15) // 16, 1)[:n_cb].contiguous()
d_bits = torch.zeros((n_cb, K // 8), dtype=torch.uint8, device=dev)
dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO)
lib.srsran_hip_dev_knob(b"SRSRAN_HIP_TDEC_LAT", lat)
for rep in range(3):
    dec.run(d_llr, pool.shape[1], d_bits, K // 8, n_cb, nit, sb, st)
torch.cuda.synchronize()
