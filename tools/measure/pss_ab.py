"""cell search (BASELINE configs[4]): the product's correlation kernel (block spectrum parked once, read back twice) against the "recompute" form
(every hypothesis transforms the block again, nothing parked); 256 captures of 10 ms, HIP events around the whole call, results compared"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import srslte_amd as S
from srslte_amd import capi
import test_gpu_sync as TS
lib = S.lib()
dev = torch.device("cuda", 0)
frame, N, prb, caps = 307200, 2048, 100, 256
rng = np.random.default_rng(5)
base = np.stack([TS._capture(c, prb, N, frame, d, 0.05, rng) for c, d in ((123, 123457), (124, 1000), (125, 250000), (360, 77777))])
d_caps = torch.from_numpy(base.view(np.float32)).to(dev).repeat(caps // 4, 1).contiguous()
d_out = torch.zeros(caps * 3 * C.sizeof(capi.HipCell), dtype=torch.uint8, device=dev)
ref = None
for variant in (b"wave", b"recompute", b"wave", b"recompute"):
    assert lib.srsran_hip_dev_knob(b"SRSRAN_HIP_PSS_VARIANT", variant) == 0
    h = C.c_void_p()
    capi.check(lib.srsran_hip_cellsearch_create(C.byref(h), frame, N, capi.CP_NORM, 1, caps), "create")
    st = torch.cuda.current_stream().cuda_stream
    ts = []
    for rep in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); capi.check(lib.srsran_hip_cellsearch_run(h, d_caps.data_ptr(), caps, 7, d_out.data_ptr(), st), "run"); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    got = d_out.cpu().numpy().tobytes()
    ref = ref or got
    ts.sort()
    print("%-10s median %.3f ms  min %.3f ms  %6.1f k captures/s  results %s" % (variant.decode(), ts[len(ts) // 2], ts[0], caps / ts[len(ts) // 2], "same" if got == ref else "DIFFER"), flush=True)
    lib.srsran_hip_cellsearch_free(h)
