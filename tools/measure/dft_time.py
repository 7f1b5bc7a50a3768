#!/usr/bin/env python3
"""time srsran_hip_dft_batch_run for transform-precoding lengths (device resident)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import srslte_amd as S
from srslte_amd import capi
lib = S.lib()
capi.check(lib.srsran_hip_set_device(0), "dev")
st = torch.cuda.current_stream().cuda_stream
for n in [int(a) for a in sys.argv[1:]] or [1200, 900, 600, 300, 144, 72, 12]:
    how = max(1, (256 << 20) // (8 * n))
    x = torch.randn((how, n, 2), device="cuda")
    y = torch.empty_like(x)
    h = C.c_void_p()
    capi.check(lib.srsran_hip_dft_batch_create(C.byref(h), n, capi.DFT_BACKWARD, False, False, True), "create")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(2):
        e0.record()
        for _ in range(5):
            capi.check(lib.srsran_hip_dft_batch_run(h, x.data_ptr(), y.data_ptr(), how, st), "run")
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("N=%5d how=%7d %.3f ms  %.2f TB/s (in+out)" % (n, how, ms, 2 * how * n * 8 / ms / 1e9), flush=True)
    lib.srsran_hip_dft_batch_free(h)
