#!/usr/bin/env python3
"""early-stop decoder launch of a transport-block call against its half-iteration count: one transport block of 13 code blocks whose soft bits are
noise (no code block ever matches its CRC), max_iterations = 1, 2, 4, 8; HIP events around srsran_hip_sch_decode (de-matching + decoder + CRC kernels).
usage: es_time.py [n_tb]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
n_tb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = S.lib(); lib.srsran_hip_set_device(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
tbs, Qm, G = 75376, 6, 100800
ncb = O.cbsegm(tbs)["C"]
rng = np.random.default_rng(1)
for llr8 in (False, True):
    e = rng.integers(-60, 60, (n_tb, G)).astype(np.int8 if llr8 else np.int16)
    d_e = torch.from_numpy(e).to(dev)
    d_soft = torch.zeros((n_tb * ncb, 18600), dtype=torch.int8 if llr8 else torch.int16, device=dev)
    d_data = torch.zeros((n_tb, tbs // 8 + 8), dtype=torch.uint8, device=dev)
    h = C.c_void_p(); capi.check(lib.srsran_hip_sch_create(C.byref(h)), "create")
    tb = (capi.HipTb * n_tb)(*[capi.HipTb(tbs, Qm, 0x100, G, i * G, i * (tbs // 8 + 8), i * ncb) for i in range(n_tb)])
    res = (capi.HipTbResult * n_tb)()
    fn = lib.srsran_hip_sch_decode_8bit if llr8 else lib.srsran_hip_sch_decode
    row = []
    for nit in (1, 2, 4, 8):
        best = 1e9
        for rep in range(6):
            crc = np.zeros(n_tb * ncb, np.uint8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            capi.check(fn(h, d_e.data_ptr(), tb, n_tb, nit, d_soft.data_ptr(), crc.ctypes.data, d_data.data_ptr(), res, st), "decode")
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append(best)
        assert res[0].avg_iterations == nit, (res[0].avg_iterations, nit)
    print("%s n_tb=%d (%d code blocks): %s ms at 1/2/4/8 half iterations, early-stop kernel, nothing ever decodes" % ("int8 " if llr8 else "int16", n_tb, n_tb * ncb, " ".join("%.3f" % v for v in row)), flush=True)
    lib.srsran_hip_sch_free(h)
