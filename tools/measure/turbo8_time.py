#!/usr/bin/env python3
"""time the 8-bit API batch decoder (srsran_tdec_run_all_8bit: AUTO -> avx8 32-sub-block window for K > 2048) against the 16-bit one,
and the reference's own 8-bit decoder on one host core"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
n_cb = int(os.environ.get("N_CB", "65520"))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
K = 6144
_, pool8 = O.turbo_llrs_8bit(K, 16, 1.0, seed=1)
_, pool16 = O.turbo_llrs(K, 16, 1.0, seed=1)
for name, pool, llr8 in (("int16 API (avx16 window)", pool16, False), ("int8 API (avx8 window)", pool8, True)):
    d_llr = torch.from_numpy(pool).to(dev).repeat((n_cb + 15) // 16, 1)[:n_cb].contiguous()
    d_bits = torch.zeros((n_cb, K // 8), dtype=torch.uint8, device=dev)
    dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO, llr8=llr8)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if llr8:
            capi.check(S.lib().srsran_hip_tdec_batch_run_8bit(dec._h, d_llr.data_ptr(), 3 * K + 12, d_bits.data_ptr(), K // 8, n_cb, 8, 0, st), "run8")
        else:
            dec.run(d_llr, 3 * K + 12, d_bits, K // 8, n_cb, 8, 0, st)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%-28s %.3f ms per %d blocks = %.1f Gbit/s" % (name, ms, n_cb, n_cb * K / ms / 1e6), flush=True)
    del dec
if O.have_ref():
    ref = C.CDLL(O.REF_LIB)
    h = C.create_string_buffer(64 * 1024)
    assert ref.srsran_tdec_init(h, K) == 0
    ref.srsran_tdec_force_not_sb(h)
    out = np.zeros(K // 8, np.uint8)
    for name, pool, f in (("reference int16", pool16, ref.srsran_tdec_run_all), ("reference int8", pool8, ref.srsran_tdec_run_all_8bit)):
        t0 = time.perf_counter()
        for i in range(32):
            f(h, O.P(pool[i % 16].copy()), O.P(out), 8, K)
        dt = time.perf_counter() - t0
        print("%-28s %.1f Mbit/s on one host core" % (name, 32 * K / dt / 1e6))
