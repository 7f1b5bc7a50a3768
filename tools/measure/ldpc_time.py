import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
n_cw = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
bg, Z = 0, 384
_, l8 = O.ldpc_llrs(bg, Z, 16, 3.0, seed=1, clip=63)
dev = torch.device("cuda", 0)
for name, typ, arr in (("int8", capi.LDPC_C, l8), ("int16", capi.LDPC_S, (l8.astype(np.int16) * 100)), ("float", capi.LDPC_F, l8.astype(np.float32))):
    d = torch.from_numpy(arr).to(dev).repeat((n_cw + 15) // 16, 1)[:n_cw].contiguous()
    out = torch.zeros((n_cw, 22 * Z), dtype=torch.uint8, device=dev)
    b = S.LdpcBatch(bg, Z, 0.8, 20, n_cw, typ)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.run(d, arr.shape[1], out, 22 * Z, n_cw, 66 * Z, None, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%-6s %8.2f ms  %7.1f Mbit/s" % (name, ms, n_cw * 22 * Z / ms / 1e3), flush=True)
