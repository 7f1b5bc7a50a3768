#!/usr/bin/env python3
"""Latency of the reference's transport-block seam (srsran_hip_decode_tb_cb = decode_tb_cb, sch.c:370) on HOST buffers in the reference's structs:
one transport block of 13 code blocks (TBS 75376, 64-QAM), fresh soft buffer per call (what srsran_softbuffer_rx_reset_tbs leaves), p50 / p99 over
--calls calls, 16- and 8-bit, at the SNR knobs of tools/bench_tti.py.  Compare: profiles/r03_tti.json (the same block through srsran_hip_sch_decode
from pinned memory)."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
SB = 18600


class SoftbufferRx(C.Structure):
    _fields_ = [("max_cb", C.c_uint32), ("max_cb_size", C.c_uint32), ("buffer_f", C.POINTER(C.c_void_p)), ("data", C.POINTER(C.c_void_p)),
                ("cb_crc", C.POINTER(C.c_bool)), ("tb_crc", C.c_bool)]


class SchHead(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("avg_iterations", C.c_float), ("llr_is_8bit", C.c_bool)]


def lte_points(lib, capi, O, calls, snrs, with_ref=True):
    """srsran_hip_decode_tb_cb (= decode_tb_cb, sch.c:370) on host buffers: TBS 75376 (13 code blocks), fresh soft buffer per call"""
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
    for snr in snrs:
        e16, payload = O.make_tb(tbs, Qm, G, 0, snr, np.random.default_rng(int(snr * 10)))
        for llr8 in (False, True):
            e = np.clip(np.round(e16 * (24.0 / np.mean(np.abs(e16)))), -127, 127).astype(np.int8) if llr8 else e16
            q = SchHead(10, 0.0, llr8)
            t, ok = [], 0
            for i in range(calls + 10):
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
            p = {"snr_knob_db": snr, "llr": "int8" if llr8 else "int16", "p50_ms": t[len(t) // 2], "p99_ms": t[int(len(t) * 0.99) - 1], "min_ms": t[0],
                 "ok": [ok, calls], "avg_half_iterations": q.avg_iterations, "payload_ok": bool(np.array_equal(data[:tbs // 8], payload[:tbs // 8]))}
            if with_ref and O.have_ref():
                # the reference's own rm_turbo / turbodecoder / crc objects (oracle/_ref) in the order of decode_tb_cb, one core
                chain = O.RefSchChain(llr8, 10)
                tr = []
                for _ in range(8):
                    soft, crc = chain.new_softbuffer(ncb), np.zeros(ncb, np.uint8)
                    soft[:] = 0
                    t0 = time.perf_counter()
                    chain.decode_tb(tbs, Qm, 0, e, soft, crc)
                    tr.append((time.perf_counter() - t0) * 1e3)
                tr.sort()
                p["reference_one_core_ms"] = tr[len(tr) // 2]
            out["points"].append(p)
            print(json.dumps(p), file=sys.stderr, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--snrs", default="30.0,8.0,6.0")
    a = ap.parse_args()
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    lib = S.lib()
    capi.check(lib.srsran_hip_set_device(0), "set_device")
    out = lte_points(lib, capi, O, a.calls, [float(x) for x in a.snrs.split(",")])
    out["nr"] = nr_points(lib, capi, O, a.calls)
    # (several worker threads at once: tools/probe/seam_threads.c -- threads of this interpreter share its lock, and a call's return waits for it)
    print(json.dumps(out, indent=1))


def nr_points(lib, capi, O, calls, sigmas=(6.0, 9.0, 10.0), with_ref=True):
    """the NR entry point srsran_hip_sch_nr_decode_tb (= srsran_dlsch_nr_decode / srsran_ulsch_nr_decode, sch_nr.c:724-749) on host buffers: one transport
    block of 67,368 bits (8 code blocks, BG1, Z = 384, E = 12672), fresh soft buffer per call, max 10 iterations with CRC early stop; beside it the
    reference's own objects (oracle/_ref: srsran_ldpc_rm_rx_c + srsran_ldpc_decoder_decode_crc_c per code block, the decoder type its dispatch picks on
    this host) in the order of sch_nr_decode on one core"""
    fn = lib.srsran_hip_sch_nr_decode_tb
    fn.restype = C.c_int
    fn.argtypes = [C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    tbs, R, mod, Nl, G = 67368, 0.67, 4, 1, 8 * 12672
    cfg = O.sch_nr_tb_info(tbs, R, mod, G, Nl, 0)
    N, ncb = 66 * cfg.Z, cfg.C
    cfg.Nref = N
    rows = [np.zeros(N + 8, np.int16) for _ in range(ncb)]
    keep = [np.zeros(N // 8 + 8, np.uint8) for _ in range(ncb)]
    flags = np.zeros(ncb, np.bool_)
    sb = SoftbufferRx(ncb, N + 8, (C.c_void_p * ncb)(*[r.ctypes.data for r in rows]), (C.c_void_p * ncb)(*[k.ctypes.data for k in keep]),
                      flags.ctypes.data_as(C.POINTER(C.c_bool)), False)
    rng = np.random.default_rng(4)
    payload = rng.integers(0, 256, tbs // 8).astype(np.uint8)
    e = O.sch_nr_encode_tb(cfg, 0, payload)
    pts = []
    for sigma in sigmas:
        llr = np.clip(np.round(16.0 * (1.0 - 2.0 * e) + sigma * rng.standard_normal(G)), -63, 63).astype(np.int8)
        out = np.zeros(tbs // 8, np.uint8)
        crc_ok, avg = C.c_bool(False), C.c_float(0)
        tb = capi.HipNrTb(R, tbs, mod, 0, Nl, G, 0, 0, 0, 0, 0)
        t, good = [], 0
        for i in range(calls + 10):
            for r in rows:
                r[:] = 0
            flags[:] = False
            t0 = time.perf_counter()
            rc = fn(0.8, 10, C.byref(tb), llr.ctypes.data, C.byref(sb), out.ctypes.data, C.byref(crc_ok), C.byref(avg))
            dt = time.perf_counter() - t0
            assert rc == 0
            if i >= 10:
                t.append(dt * 1e3)
                good += int(crc_ok.value)
        t.sort()
        p = {"noise_knob": sigma, "p50_ms": t[len(t) // 2], "p99_ms": t[int(len(t) * 0.99) - 1], "ok": [good, calls], "avg_iterations": avg.value,
             "payload_ok": bool(np.array_equal(out, payload))}
        if with_ref and O.have_ref():
            p["reference_one_core"] = ref_nr_tb(O, cfg, llr, 10)
        pts.append(p)
        print(json.dumps(p), file=sys.stderr, flush=True)
    return {"what": "srsran_hip_sch_nr_decode_tb, TBS %d (%d code blocks BG1 Z=%d), host buffers, fresh soft buffer per call, max 10 iterations" % (tbs, ncb, cfg.Z),
            "points": pts}


def ref_nr_tb(O, cfg, llr, max_iter, reps=12):
    class Args(C.Structure):
        _fields_ = [("type", C.c_int), ("bg", C.c_int), ("ls", C.c_uint16), ("scaling_fctr", C.c_float), ("max_nof_iter", C.c_uint32)]

    flags = open("/proc/cpuinfo").read()
    lib512 = os.path.join(os.path.dirname(O.REF_LIB), "libsrsran_ref_avx512.so")
    use512 = " avx512f" in flags and " avx512bw" in flags and os.path.exists(lib512)
    ref = C.CDLL(lib512 if use512 else O.REF_LIB)
    dec, rmr, crc = C.create_string_buffer(4096), C.create_string_buffer(64), C.create_string_buffer(4096)
    assert ref.srsran_ldpc_decoder_init(dec, C.byref(Args(6 if use512 else 4, cfg.bg, cfg.Z, 0.8, max_iter))) == 0
    assert ref.srsran_ldpc_rm_rx_init_c(rmr) == 0 and ref.srsran_crc_init(crc, C.c_uint32(0x1800063), 24) == 0
    N, Kr = 66 * cfg.Z, 22 * cfg.Z
    E = [O.sch_nr_get_E(cfg, r) for r in range(cfg.C)]
    temp = np.zeros(Kr, np.uint8)
    t, its = [], 0
    for _ in range(reps):
        soft = np.zeros((cfg.C, N), np.int8)
        packed = np.zeros((cfg.C, Kr // 8), np.uint8)
        t0 = time.perf_counter()
        inp, its = 0, 0
        for r in range(cfg.C):
            n_llr = ref.srsran_ldpc_rm_rx_c(rmr, O.P(llr[inp:inp + E[r]]), O.P(soft[r]), C.c_uint32(E[r]), C.c_uint32(cfg.F), C.c_int(cfg.bg),
                                            C.c_uint32(cfg.Z), C.c_uint8(0), C.c_int(4), C.c_uint32(N))
            ret = ref.srsran_ldpc_decoder_decode_crc_c(dec, O.P(soft[r]), O.P(temp), C.c_uint32(n_llr), crc)
            its += max_iter if ret == 0 else ret
            ref.srsran_bit_pack_vector(O.P(temp), O.P(packed[r]), C.c_int(cfg.Kp - cfg.L_cb))
            inp += E[r]
        t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    return {"ms_per_tb_median": t[len(t) // 2], "avg_iterations": its / cfg.C, "decoder_type": "C_AVX512" if use512 else "C_AVX2"}


if __name__ == "__main__":
    main()
