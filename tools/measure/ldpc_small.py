"""small LDPC batches: check-to-variable messages in LDS (default for <= 256 code words) against the global slabs (SRSRAN_HIP_LDPC_C2V_LDS=0);
BG1 / BG2, several lifting sizes, device resident, HIP events around one launch"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
lib = S.lib()
dev = torch.device("cuda", 0)
for bg, Z, iters in ((0, 384, 20), (0, 384, 4), (0, 256, 20), (1, 384, 20), (0, 96, 20)):
    K = (22 if bg == 0 else 10) * Z
    N = (66 if bg == 0 else 50) * Z
    _, l8 = O.ldpc_llrs(bg, Z, 16, 3.0, seed=1, clip=63)
    ref, _ = O.ldpc_decode(bg, Z, l8[:2], 0.8, iters)
    for n_cw in (1, 8, 64, 256):
        row = []
        for knob in (b"0", b"1"):
            assert lib.srsran_hip_dev_knob(b"SRSRAN_HIP_LDPC_C2V_LDS", knob) == 0
            d = torch.from_numpy(l8).to(dev).repeat((n_cw + 15) // 16, 1)[:n_cw].contiguous()
            out = torch.zeros((n_cw, K), dtype=torch.uint8, device=dev)
            b = S.LdpcBatch(bg, Z, 0.8, iters, n_cw, capi.LDPC_C)
            best = 1e9
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); b.run(d, l8.shape[1], out, K, n_cw, N, None, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            ok = np.array_equal(out[:min(2, n_cw)].cpu().numpy(), ref[:min(2, n_cw)])
            row.append("%s %7.1f us%s" % ("lds " if knob == b"1" else "slab", best * 1e3, "" if ok else " MISMATCH"))
        print("BG%d Z=%3d it=%2d n_cw=%3d   %s   %s" % (bg + 1, Z, iters, n_cw, row[0], row[1]), flush=True)
