# srsran_hip_sch_decode on 2496 transport blocks (13 code blocks of 6144 each): one caller against W worker threads with a handle, a stream and a
# share of the blocks each (the reference's worker pool); wall time per step over all workers
import sys, time, threading, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
lib = S.lib(); dev = torch.device("cuda", 0)
capi.check(lib.srsran_hip_set_device(0), "set_device")
n_tb, tbs, Qm, G, snr, iters = 2496, 75376, 6, 100800, 12.0 if len(sys.argv) < 2 else float(sys.argv[1]), 8
ncb = O.cbsegm(tbs)["C"]
rng = np.random.default_rng(4)
pool = [O.make_tb(tbs, Qm, G, 0, snr, rng) for _ in range(8)]
e_pool = torch.from_numpy(np.stack([p[0] for p in pool])).to(dev)
d_e = e_pool.repeat(n_tb // 8, 1).contiguous()
dlen = tbs // 8 + 8
d_data = torch.zeros((n_tb, dlen), dtype=torch.uint8, device=dev)
d_soft = torch.zeros((n_tb * ncb, capi.SOFTBUFFER_CB_SIZE), dtype=torch.int16, device=dev)
for W in (1, 2, 3, 4):
    m = n_tb // W
    hs, sts, arrs, ress, flgs = [], [], [], [], []
    for w in range(W):
        h = C.c_void_p(); capi.check(lib.srsran_hip_sch_create(C.byref(h)), "create"); hs.append(h)
        sts.append(torch.cuda.Stream())
        arrs.append((capi.HipTb * m)(*[capi.HipTb(tbs, Qm, 0x100, G, (w * m + i) * G, (w * m + i) * dlen, (w * m + i) * ncb) for i in range(m)]))
        ress.append((capi.HipTbResult * m)()); flgs.append(np.zeros(n_tb * ncb, np.uint8))
    bar = threading.Barrier(W + 1)
    stop = False
    def work(w):
        while True:
            bar.wait()
            if stop:
                return
            flgs[w][:] = 0
            capi.check(lib.srsran_hip_sch_decode(hs[w], d_e.data_ptr(), arrs[w], m, iters, d_soft.data_ptr(), flgs[w].ctypes.data, d_data.data_ptr(), ress[w],
                                                 sts[w].cuda_stream), "decode")
            bar.wait()
    ths = [threading.Thread(target=work, args=(w,)) for w in range(W)]
    [t.start() for t in ths]
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); bar.wait(); bar.wait(); ts.append(time.perf_counter() - t0)
    stop = True; bar.wait(); [t.join() for t in ths]
    ok = sum(1 for r in ress for x in r if x.crc_ok == 0)
    dt = min(ts[1:])
    print("%d worker(s): %.3f ms per %d blocks = %.1f Gbit/s of TBS, crc ok %d" % (W, dt * 1e3, W * m, W * m * tbs / dt / 1e9, ok), flush=True)
