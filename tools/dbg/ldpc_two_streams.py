# LDPC int8 batch (BG1 Z=384, 20 iterations): 16,384 words as ONE launch against two half-size launches on two streams at once
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
n_cw, bg, Z = 16384, 0, 384
_, l8 = O.ldpc_llrs(bg, Z, 16, 3.0, seed=1, clip=63)
dev = torch.device("cuda", 0)
d = torch.from_numpy(l8).to(dev).repeat(n_cw // 16, 1).contiguous()
out = torch.zeros((n_cw, 22 * Z), dtype=torch.uint8, device=dev)
def timed(fn):
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
one = S.LdpcBatch(bg, Z, 0.8, 20, n_cw, capi.LDPC_C)
cur = torch.cuda.current_stream()
t1 = timed(lambda: one.run(d, l8.shape[1], out, 22 * Z, n_cw, 66 * Z, None, cur.cuda_stream))
print("one launch of %d words           : %7.2f ms  %7.1f Mbit/s" % (n_cw, t1, n_cw * 22 * Z / t1 / 1e3), flush=True)
for parts in (2, 4):
    m = n_cw // parts
    objs = [S.LdpcBatch(bg, Z, 0.8, 20, m, capi.LDPC_C) for _ in range(parts)]
    sts = [torch.cuda.Stream() for _ in range(parts)]
    def go():
        for k in range(parts):
            sts[k].wait_stream(cur)
            objs[k].run(d[k * m:(k + 1) * m], l8.shape[1], out[k * m:(k + 1) * m], 22 * Z, m, 66 * Z, None, sts[k].cuda_stream)
        for k in range(parts):
            cur.wait_stream(sts[k])
    t = timed(go)
    print("%d launches of %d words, %d streams: %7.2f ms  %7.1f Mbit/s" % (parts, m, parts, t, n_cw * 22 * Z / t / 1e3), flush=True)
