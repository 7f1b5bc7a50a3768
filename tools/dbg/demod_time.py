#!/usr/bin/env python3
"""dev timing of the demod/descramble kernel: batched with / without scrambling, and one huge job"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import srslte_amd as S
from srslte_amd import capi
lib = S.lib()
dev = torch.device("cuda", 0)
ntb, nsym = 2048, 16800
mod = int(sys.argv[1]) if len(sys.argv) > 1 else 3
typ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Qm = [1, 2, 4, 6, 8][mod]
esz = [2, 1, 4][typ]
d_sym = torch.randn((ntb * nsym, 2), dtype=torch.float32, device=dev)
d_e = torch.zeros(ntb * nsym * Qm * esz, dtype=torch.uint8, device=dev)
h = C.c_void_p()
capi.check(lib.srsran_hip_demod_create(C.byref(h)), "create")
st = torch.cuda.current_stream().cuda_stream
def run(jobs, n, label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(5):
        ev[0].record()
        capi.check(lib.srsran_hip_demod_run(h, d_sym.data_ptr(), d_e.data_ptr(), typ, jobs, n, st), "run")
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]))
    by = ntb * nsym * (8 + Qm * esz)
    print("%-28s %.3f ms  %.0f GB/s" % (label, best, by / best / 1e6))
for scr in (1, 0):
    jobs = (capi.HipDemodJob * ntb)(*[capi.HipDemodJob(mod, nsym, i * nsym, i * nsym * Qm, 1000 + i, scr) for i in range(ntb)])
    run(jobs, ntb, "2048 jobs scramble=%d" % scr)
big = (capi.HipDemodJob * 1)(capi.HipDemodJob(mod, ntb * nsym, 0, 0, 0, 0))
run(big, 1, "1 job no scramble")
per = (1 << 21) // Qm
nj = (ntb * nsym) // per
jobs = (capi.HipDemodJob * nj)(*[capi.HipDemodJob(mod, per, i * per, i * per * Qm, 1000 + i, 1) for i in range(nj)])
run(jobs, nj, "%d long jobs scramble=1 (%.0f%% of the data)" % (nj, 100.0 * nj * per / (ntb * nsym)))
