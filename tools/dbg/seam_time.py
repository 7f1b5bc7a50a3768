#!/usr/bin/env python3
"""Latency of the reference's transport-block seam (srsran_hip_decode_tb_cb = decode_tb_cb, sch.c:370) on HOST buffers in the reference's structs:
one transport block of 13 code blocks (TBS 75376, 64-QAM), fresh soft buffer per call (what srsran_softbuffer_rx_reset_tbs leaves), p50 / p99 over
--calls calls, 16- and 8-bit, at the SNR knobs of tools/bench_tti.py.  Compare: profiles/r03_tti.json (the same block through srsran_hip_sch_decode
from pinned memory)."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
SB = 18600


class SoftbufferRx(C.Structure):
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_f", C.POINTER(C.c_void_p)), ("data", C.POINTER(C.c_void_p)),
                ("cb_crc", C.POINTER(C.c_bool)), ("tb_crc", C.c_bool)]


class SchHead(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("avg_iterations", C.c_float), ("llr_is_8bit", C.c_bool)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--snrs", default="30.0,8.0,6.0")
    a = ap.parse_args()
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    lib = S.lib()
    capi.check(lib.srsran_hip_set_device(0), "set_device")
    fn = lib.srsran_hip_decode_tb_cb
    fn.restype = C.c_bool
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    tbs, Qm, G = 75376, 6, 100800
    seg = O.cbsegm(tbs)
    ncb = seg["C"]
    cs = capi.Cbsegm()
    assert lib.srsran_cbsegm(C.byref(cs), tbs) == 0
    rows = [np.zeros(SB, np.int16) for _ in range(ncb)]
    keep = [np.zeros(SB // 8, np.uint8) for _ in range(ncb)]
    flags = np.zeros(ncb, np.bool_)
    sb = SoftbufferRx(ncb, SB, (C.c_void_p * ncb)(*[r.ctypes.data for r in rows]), (C.c_void_p * ncb)(*[k.ctypes.data for k in keep]),
                      flags.ctypes.data_as(C.POINTER(C.c_bool)), False)
    data = np.zeros(tbs // 8 + 8, np.uint8)
    out = {"what": "srsran_hip_decode_tb_cb, TBS %d (%d code blocks), host buffers, fresh soft buffer per call, max 10 half iterations" % (tbs, ncb), "points": []}
    for snr in [float(x) for x in a.snrs.split(",")]:
        e16, payload = O.make_tb(tbs, Qm, G, 0, snr, np.random.default_rng(int(snr * 10)))
        for llr8 in (False, True):
            e = np.clip(np.round(e16 * (24.0 / np.mean(np.abs(e16)))), -127, 127).astype(np.int8) if llr8 else e16
            q = SchHead(10, 0.0, llr8)
            t, ok = [], 0
            for i in range(a.calls + 10):
                for r in rows:  # srsran_softbuffer_rx_reset_tbs (softbuffer.c:147-167), outside the timed region as in pusch_test.c
                    r[:] = 0
                flags[:] = False
                t0 = time.perf_counter()
                good = fn(C.byref(q), C.byref(sb), C.byref(cs), Qm, 0, G, e.ctypes.data, data.ctypes.data)
                dt = time.perf_counter() - t0
                if i >= 10:
                    t.append(dt * 1e3)
                    ok += int(good)
            t.sort()
            out["points"].append({"snr_knob_db": snr, "llr": "int8" if llr8 else "int16", "p50_ms": t[len(t) // 2], "p99_ms": t[int(len(t) * 0.99) - 1], "min_ms": t[0],
                                  "ok": [ok, a.calls], "avg_half_iterations": q.avg_iterations, "payload_ok": bool(np.array_equal(data[:tbs // 8], payload[:tbs // 8]))})
            print(json.dumps(out["points"][-1]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


main()
