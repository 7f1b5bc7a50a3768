#!/usr/bin/env python3
"""time srsran_hip_ofdm_batch_{rx,tx} for the LTE 20 MHz and NR 100 MHz shapes (device resident)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import srslte_amd as S
st = torch.cuda.current_stream().cuda_stream
for prb, nfft, n_sf in ((100, 2048, 4096), (100, 1536, 4096), (273, 4096, 2048), (6, 128, 65536)):
    for tx in (False, True):
        o = S.OfdmBatch(prb, tx=tx, symbol_sz=nfft, normalize=True)
        n_in, n_out = (o.sf_re, o.sf_sz) if tx else (o.sf_sz, o.sf_re)
        x = torch.randn((n_sf, n_in, 2), device="cuda")
        y = torch.zeros((n_sf, n_out, 2), device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(2):
            e0.record()
            for _ in range(5):
                o.run(x.data_ptr(), y.data_ptr(), n_sf, st)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%3d PRB N=%4d %s  %6d sf  %.3f ms  %.2f TB/s (in+out)  %.0f Msamples/s" % (prb, nfft, "tx" if tx else "rx", n_sf, ms, n_sf * (n_in + n_out) * 8 / ms / 1e9,
                                                                                     n_sf * o.sf_sz / ms / 1e3), flush=True)
