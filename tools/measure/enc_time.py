#!/usr/bin/env python3
"""latency of the transmit-side entry srsran_hip_encode_tb (encode_tb, sch.c:239) on host buffers: p50 of 200 calls per transport block size"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srslte_amd as S, oracle_api as O
from srslte_amd import capi


class SoftbufferTx(C.Structure):
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_b", C.POINTER(C.c_void_p))]


lib = S.lib(); lib.srsran_hip_set_device(0)
fn = lib.srsran_hip_encode_tb
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rows = [np.zeros(18600, np.uint8) for _ in range(14)]
sb = SoftbufferTx(14, 18600, (C.c_void_p * 14)(*[r.ctypes.data for r in rows]))
rng = np.random.default_rng(1)
for tbs, Qm, G in ((75376, 6, 86400), (15264, 4, 14400), (4584, 2, 9000), (936, 2, 2400)):
    cs = capi.Cbsegm()
    assert lib.srsran_cbsegm(C.byref(cs), tbs) == 0
    data = rng.integers(0, 256, tbs // 8 + 8).astype(np.uint8)
    out = np.zeros(G // 8 + 8, np.uint8)
    t = []
    for i in range(220):
        t0 = time.perf_counter()
        assert fn(C.byref(sb), C.byref(cs), Qm, 0, G, data.ctypes.data, out.ctypes.data) == 0
        if i >= 20:
            t.append((time.perf_counter() - t0) * 1e6)
    t.sort()
    e = O.tb_coded_bits(tbs, Qm, G, 0, None, payload=np.unpackbits(data[:tbs // 8]), tx_order=True)[0]
    print("TBS %6d (%2d code blocks), G %6d: p50 %.1f us  p99 %.1f us   equal to the oracle's chain: %s" % (tbs, cs.C, G, t[len(t) // 2], t[int(len(t) * 0.99) - 1], np.array_equal(np.unpackbits(out)[:e.size], e)), flush=True)
