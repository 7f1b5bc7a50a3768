# time the int8 LDPC batch (BG1 Z=384, 20 iterations, 16384 words) with the library named by SRSRAN_HIP_LIB; timing only (variants may be wrong on purpose)
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
n_cw = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
bg, Z = 0, int(sys.argv[2]) if len(sys.argv) > 2 else 384
_, l8 = O.ldpc_llrs(bg, Z, 16, 3.0, seed=1, clip=63)
dev = torch.device("cuda", 0)
d = torch.from_numpy(l8).to(dev).repeat((n_cw + 15) // 16, 1)[:n_cw].contiguous()
out = torch.zeros((n_cw, 22 * Z), dtype=torch.uint8, device=dev)
b = S.LdpcBatch(bg, Z, 0.8, 20, n_cw, capi.LDPC_C)
best = 1e9
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.run(d, l8.shape[1], out, 22 * Z, n_cw, 66 * Z, None, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("%8.2f ms  %7.1f Mbit/s" % (best, n_cw * 22 * Z / best / 1e3), flush=True)
