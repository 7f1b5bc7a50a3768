"""The single-GPU configurations of BASELINE.json other than the headline, as legs of bench.py (`extra.<name>` in its JSON line):

    leg_ldpc        configs[2]  NR 100 MHz: srsran_ofdm_rx_sf N=4096 / 273 PRB + LDPC BG1 Z=384 int8 layered min-sum, 20 iterations
    leg_cellsearch  configs[4]  PSS (3 N_id_2 hypotheses) + SSS over 10 ms captures at 30.72 Msps -> 504 PCI hypotheses
    leg_uplink      configs[3]  per-GPU shard of the multi-UE PUSCH receive chain (time samples -> transport blocks)

Every leg: inputs made by the library's own transmit side (encoders, rate matchers, OFDM modulator, PSS / SSS generators, scrambling
sequences; the 64-QAM mapper is the one thing synthesised here -- the product has no modulator), resident in HBM before the timed
region; the oracle is only touched inside the `cpu_baseline` blocks; barrier + synchronize on both sides of `steps` timed steps, max over ranks; kernel time by HIP events on the launch
stream; a `roofline` against the algorithmic bytes of SURVEY.md par. 8(d) and -- on rank 0 of a 1-GPU run -- a `cpu_baseline`
on ONE host core over a bounded sample, which is also the leg's parity check.  The CPU side uses the reference's own compiled
decoder (oracle/_ref) when present, else the C / scipy restatement ("port")."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0


def load_traffic():
    """HBM traffic per launch from the PMC passes of the round (profiles/rNN_traffic.json, newest round first)"""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)
        except Exception:
            continue
    return None


def traffic_of(tj, kernel, units, units_key):
    """(bytes per launch scaled to `units`, source note) or (None, None)"""
    try:
        k = tj[kernel]
        return k["traffic_bytes_per_launch"] * units / float(k[units_key]), tj.get("source")
    except Exception:
        return None, None


def _P(a):
    return a.ctypes.data_as(C.c_void_p)


def _scrambling_bits(lib, seed, n):
    """c(n) of the Gold sequence of `seed` (36.211 7.2) from the library's own descrambler: srsran_sequence_apply_c flips the sign where c = 1"""
    ones, out = np.ones(n, np.int8), np.zeros(n, np.int8)
    lib.srsran_sequence_apply_c(_P(ones), _P(out), n, seed)
    return (out < 0).astype(np.uint8)


def _qam64(bits):
    """36.211 Table 7.1.4-1 (bench input synthesis only: the product has no modulator, it is outside the hot path)"""
    b = 1.0 - 2.0 * np.asarray(bits, np.float64).reshape(-1, 6)
    i = b[:, 0] * (4.0 - b[:, 2] * (2.0 - b[:, 4]))
    q = b[:, 1] * (4.0 - b[:, 3] * (2.0 - b[:, 5]))
    return ((i + 1j * q) / np.sqrt(42.0)).astype(np.complex64)


LAST_TIMING = {}


def _timed(ctx, torch, step, steps, warmup):
    """the bench contract for one leg: warm-up, barrier + synchronize, `steps` steps, synchronize + barrier; returns wall seconds of this rank.

    Attribution of the timed region (no extra synchronisation: one more event per step on the launch stream, host clocks around the phases)
    is left in LAST_TIMING -- this rank's figures, milliseconds:
      step_ms_{min,median,max}  deltas of consecutive per-step end events (the first one against an event recorded before step 0)
      gpu_span_ms               first start event -> last end event
      submit_ms                 host time to submit all steps (a host that cannot keep ahead of the device shows here)
      drain_ms                  host wait in the synchronize after the last submit
      barrier_ms                wall of the closing barrier + synchronize (the collective of the contract; at world size 1 it is pure overhead)
      outside_gpu_ms            wall - gpu_span: everything of the timed region that is not the device running the steps
      slowest_step              index of the step with the largest delta"""
    for _ in range(warmup):
        step(None)
    torch.cuda.synchronize()
    # the closing barrier's collective is warmed here, outside the timed region (its first use builds / launches what later ones reuse)
    ctx.barrier()
    torch.cuda.synchronize()
    ctx.barrier()
    torch.cuda.synchronize()
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ends[0].record()
    for i in range(steps):
        step(i)
        ends[i + 1].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ctx.barrier()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    d = [ends[i].elapsed_time(ends[i + 1]) for i in range(steps)]
    srt = sorted(d)
    span = ends[0].elapsed_time(ends[steps])
    LAST_TIMING.clear()
    LAST_TIMING.update({"step_ms_min": srt[0], "step_ms_median": srt[len(srt) // 2], "step_ms_max": srt[-1], "slowest_step": d.index(srt[-1]),
                        "gpu_span_ms": span, "submit_ms": (t1 - t0) * 1e3, "drain_ms": (t2 - t1) * 1e3, "barrier_ms": (t3 - t2) * 1e3,
                        "outside_gpu_ms": (t3 - t0) * 1e3 - span})
    return t3 - t0


def timing_fields():
    return {k: round(v, 4) if isinstance(v, float) else v for k, v in LAST_TIMING.items()}


def _host_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            return (" %s " % flag) in f.read().replace("\n", " ")
    except OSError:
        return False


def _host_has_avx2():
    return _host_has("avx2")


# ------------------------------------------------------------------------------------------------ configs[2]: NR LDPC + OFDM 4096

class _RefLdpcArgs(C.Structure):  # srsran_ldpc_decoder_args_t, ldpc_decoder.h:58-64
    _fields_ = [("type", C.c_int), ("bg", C.c_int), ("ls", C.c_uint16), ("scaling_fctr", C.c_float), ("max_nof_iter", C.c_uint32)]


def ldpc_cpu_baseline(bg, Z, llrs, iters, sf=0.8, budget_s=4.0):
    """the reference's srsran_ldpc_decoder_decode_c, SRSRAN_LDPC_DECODER_C_AVX2 (ldpc_dec_c_avx2long.c for Z=384), one core,
    fixed `iters` iterations -- or the scalar C restatement when oracle/_ref or AVX2 is missing.  Returns (messages, info)."""
    import oracle_api as O

    K = (22 if bg == 0 else 10) * Z
    n = llrs.shape[0]
    if O.have_ref() and _host_has_avx2():
        # the decoder type the reference's own dispatch would pick on this host (ldpc_decoder.c:609-646): AVX-512 where the host has it
        lib512 = os.path.join(os.path.dirname(O.REF_LIB), "libsrsran_ref_avx512.so")
        use512 = _host_has("avx512f") and _host_has("avx512bw") and os.path.exists(lib512)
        ref = C.CDLL(lib512 if use512 else O.REF_LIB)
        tname = "C_AVX512" if use512 else "C_AVX2"
        dec = C.create_string_buffer(4096)
        args = _RefLdpcArgs(6 if use512 else 4, bg, Z, sf, iters)  # SRSRAN_LDPC_DECODER_C_AVX512 / C_AVX2 (ldpc_decoder.h:41-53)
        assert ref.srsran_ldpc_decoder_init(dec, C.byref(args)) == 0
        out = np.zeros((n, K), np.uint8)
        t0 = time.perf_counter()
        done = 0
        for i in range(n):
            ref.srsran_ldpc_decoder_decode_c(dec, O.P(llrs[i]), O.P(out[i]), llrs.shape[1])
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        # ... and around the same words again until the sample is about two seconds of CPU work
        again, scratch = 0, np.zeros(K, np.uint8)
        while done == n and time.perf_counter() - t0 < 2.0:
            ref.srsran_ldpc_decoder_decode_c(dec, O.P(llrs[again % n]), O.P(scratch), llrs.shape[1])
            again += 1
        dt = time.perf_counter() - t0
        ref.srsran_ldpc_decoder_free(dec)
        return out[:done], {"value": (done + again) * K / dt / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "reference",
                            "sample": "%d code words BG%d Z=%d (%d distinct, compared with the device), %d iterations, reference "
                                      "srsran_ldpc_decoder_decode_c type %s (oracle/_ref; the type its own dispatch picks on this host), %.1f s on a single thread"
                                      % (done + again, bg + 1, Z, done, iters, tname, dt)}
    m = min(n, 8)
    t0 = time.perf_counter()
    out, _ = O.ldpc_decode(bg, Z, llrs[:m], sf, iters)
    dt = time.perf_counter() - t0
    return out, {"value": m * K / dt / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "port",
                 "sample": "%d code words BG%d Z=%d, %d iterations, scalar C restatement (oracle/), single thread" % (m, bg + 1, Z, iters)}


def leg_ldpc(ctx, steps=3, warmup=1, want_cpu=True, cw=16384, slots=2048, iters=20):
    import torch

    import srslte_amd as S
    from srslte_amd import capi

    lib, dev, st = S.lib(), ctx.dev, ctx.stream
    bg, Z = 0, 384
    K, N = 22 * Z, 66 * Z
    # code words from the library's own encoder, BPSK over AWGN at 3 dB (ldpc_chain_test's operating point), int8 LLRs clipped to +-63
    pool_n = 32
    g = torch.Generator(device=dev)
    g.manual_seed(11 + ctx.rank)
    msgs = torch.randint(0, 2, (pool_n, K), generator=g, device=dev, dtype=torch.uint8)
    d_cw = torch.zeros((pool_n, N), dtype=torch.uint8, device=dev)
    h = C.c_void_p()
    capi.check(lib.srsran_hip_nr_sch_create(C.byref(h)), "nr_sch_create")
    jobs = (capi.HipLdpcCb * pool_n)(*[capi.HipLdpcCb(i * K, i * N, N) for i in range(pool_n)])
    capi.check(lib.srsran_hip_ldpc_encode_batch(h, msgs.data_ptr(), d_cw.data_ptr(), jobs, pool_n, bg, Z, st), "ldpc_encode_batch")
    torch.cuda.synchronize()
    sigma = 10 ** (-3.0 / 20)
    y = 1.0 - 2.0 * d_cw.float() + sigma * torch.randn((pool_n, N), generator=g, device=dev)
    pool = torch.clamp(torch.round(y * (2.0 / sigma ** 2 * 4)), -63, 63).to(torch.int8)
    reps = (cw + pool_n - 1) // pool_n
    d_llr = pool.repeat(reps, 1)[:cw].contiguous()
    d_msg = torch.zeros((cw, K), dtype=torch.uint8, device=dev)
    dec = S.LdpcBatch(bg, Z, 0.8, iters, cw)
    ofdm = S.OfdmBatch(273, False, 4096, normalize=True, keep_dc=True)
    d_time = torch.view_as_complex(torch.randn((slots, ofdm.sf_sz, 2), generator=g, device=dev) * 0.7071)
    d_re = torch.zeros((slots, ofdm.sf_re), dtype=torch.complex64, device=dev)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

    def step(i):
        ev = evs[i] if i is not None else None
        if ev:
            ev[0].record()
        ofdm.run(d_time, d_re, slots, st)
        if ev:
            ev[1].record()
        dec.run(d_llr, N, d_msg, K, cw, N, None, st)
        if ev:
            ev[2].record()

    dt = _timed(ctx, torch, step, steps, warmup)
    t_o = sum(e[0].elapsed_time(e[1]) for e in evs) / steps * 1e-3
    t_l = sum(e[1].elapsed_time(e[2]) for e in evs) / steps * 1e-3
    dt, t_o, t_l = ctx.max_over_ranks([dt, t_o, t_l])
    if ctx.rank != 0:
        return None
    cw_bytes = N + K  # 25,344 B of LLRs in + 8,448 B of message out
    sf_bytes = 8 * (15 * 4096 + 14 * 12 * 273)
    tj = load_traffic()
    tr_l, src = traffic_of(tj, "ldpc_packed_kernel", cw, "code_words_per_launch")
    tr_o, _ = traffic_of(tj, "ofdm_kernel_n4096", slots, "slots_per_launch")
    out = {"metric": "LDPC decoded Mbit/s (info bits, NR BG1 Z=384, %d iterations) incl. OFDM demod N=4096 of %d slots" % (iters, slots),
           "value": ctx.world * cw * K * steps / dt / 1e6, "unit": "Mbit/s", "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "timing": timing_fields(), "dtype": "int8", "scaling": "weak",
           "config": {"workload": "NR 100 MHz SCS 30 kHz (BASELINE configs[2]): ofdm_rx N=4096 273 PRB x %d slots + ldpc BG1 Z=384 x %d code words, "
                                  "%d iterations, scaling 0.8, no early stop; %d distinct code words (device encoder + AWGN 3 dB) tiled %dx, per GPU"
                                  % (slots, cw, iters, pool_n, reps)},
           "ldpc_kernel_mbit_per_s": ctx.world * cw * K / t_l / 1e6, "ofdm_msamples_per_s": ctx.world * slots * ofdm.sf_sz / t_o / 1e6,
           "roofline": {"kernel": "ldpc_packed_kernel<false>", "bound": "hbm", "achieved": cw * cw_bytes / t_l / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": cw * cw_bytes / t_l / 1e9 / HBM_PEAK_GBS, "traffic": tr_l, "traffic_source": src,
                        "avg_launch_ms": t_l * 1e3, "algorithmic_bytes_per_launch": cw * cw_bytes,
                        "note": "20 iterations over on-chip soft bits + a check-to-variable store: the honest bound is VALU issue (DESIGN.md par. 3.3)"},
           "roofline_ofdm": {"kernel": "ofdm_kernel<Plan<4096,...>,rx>", "bound": "hbm", "achieved": slots * sf_bytes / t_o / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": slots * sf_bytes / t_o / 1e9 / HBM_PEAK_GBS, "traffic": tr_o, "traffic_source": src,
                             "avg_launch_ms": t_o * 1e3, "algorithmic_bytes_per_launch": slots * sf_bytes}}
    if want_cpu:
        host = pool.cpu().numpy()
        ref_out, info = ldpc_cpu_baseline(bg, Z, host, iters)
        got = d_msg[:pool_n].cpu().numpy()[:ref_out.shape[0]]
        info["parity_vs_gpu"] = "bit-exact" if np.array_equal(ref_out, got) else "MISMATCH"
        out["cpu_baseline"] = info
        out["speedup_vs_cpu_baseline"] = out["value"] / info["value"]
        sent = msgs.cpu().numpy()[:ref_out.shape[0]]
        out["blocks_recovered"] = [int((got == sent).all(axis=1).sum()), int(ref_out.shape[0])]
    del dec
    lib.srsran_hip_nr_sch_free(h)
    return out


# ------------------------------------------------------------------------------------------------ configs[4]: cell search

def leg_cellsearch(ctx, steps=3, warmup=1, want_cpu=True, caps=256):
    import torch

    import srslte_amd as S
    from srslte_amd import capi

    lib, dev, st = S.lib(), ctx.dev, ctx.stream
    frame, N = 307200, 2048
    # four captures: noise + one cell each (PSS / SSS subframe by the library's own modulator), known delays; tiled
    cells = ((11, 5000), (250, 200000), (503, 123457), (300, 77777))
    otx = S.OfdmBatch(100, tx=True, symbol_sz=N, normalize=True)
    grids = np.zeros((len(cells), 14, 1200), np.complex64)
    k0 = 600 - 31
    for i, (cid, _) in enumerate(cells):
        zc, s0, s5 = np.zeros(62, np.complex64), np.zeros(62, np.float32), np.zeros(62, np.float32)
        assert lib.srsran_pss_generate(_P(zc), cid % 3) == 0
        lib.srsran_sss_generate(_P(s0), _P(s5), cid)
        grids[i, 6, k0:k0 + 62] = zc      # last symbol of slot 0 (sync_test.c:130-150)
        grids[i, 5, k0:k0 + 62] = s0
    d_grid = torch.from_numpy(grids.reshape(len(cells), -1).view(np.float32)).to(dev)
    d_sf = torch.zeros((len(cells), otx.sf_sz, 2), dtype=torch.float32, device=dev)
    otx.run(d_grid.data_ptr(), d_sf.data_ptr(), len(cells), st)
    g = torch.Generator(device=dev)
    g.manual_seed(77 + ctx.rank)
    base = torch.randn((len(cells), frame, 2), generator=g, device=dev) * 0.05
    torch.cuda.synchronize()
    for i, (_, d) in enumerate(cells):
        base[i, d:d + otx.sf_sz] += d_sf[i]
    d_caps = base.repeat((caps + 3) // 4, 1, 1)[:caps].contiguous()
    h = C.c_void_p()
    capi.check(lib.srsran_hip_cellsearch_create(C.byref(h), frame, N, capi.CP_NORM, 1, caps), "cellsearch_create")
    d_cells = torch.zeros(caps * 3 * C.sizeof(capi.HipCell), dtype=torch.uint8, device=dev)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]

    def step(i):
        if i is not None:
            evs[i][0].record()
        capi.check(lib.srsran_hip_cellsearch_run(h, d_caps.data_ptr(), caps, 7, d_cells.data_ptr(), st), "cellsearch_run")
        if i is not None:
            evs[i][1].record()

    dt = _timed(ctx, torch, step, steps, warmup)
    t_k = sum(e[0].elapsed_time(e[1]) for e in evs) / steps * 1e-3
    dt, t_k = ctx.max_over_ranks([dt, t_k])
    got = (capi.HipCell * (caps * 3)).from_buffer_copy(d_cells.cpu().numpy().tobytes())
    ok = all(got[i * 3 + cid % 3].N_id_1 == cid // 3 and got[i * 3 + cid % 3].peak_pos == d + 15 * N // 2 and got[i * 3 + cid % 3].sf_idx == 0
             for i, (cid, d) in enumerate(cells))
    ok = ok and all(got[(i + 4) * 3 + n2].peak_pos == got[i * 3 + n2].peak_pos for i in range(min(4, caps - 4)) for n2 in range(3))
    lib.srsran_hip_cellsearch_free(h)
    if ctx.rank != 0:
        return None
    cap_bytes = frame * 8
    tj = load_traffic()
    tr, src = traffic_of(tj, "pss_wave_kernel", caps, "captures_per_launch")
    out = {"metric": "cell search Msamples/s (10 ms captures at 30.72 Msps, 3 PSS hypotheses + SSS = 504 PCI hypotheses per capture)",
           "value": ctx.world * caps * frame * steps / dt / 1e6, "unit": "Msamples/s", "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "timing": timing_fields(), "dtype": "f32", "scaling": "weak",
           "config": {"workload": "BASELINE configs[4]: %d captures x 307,200 samples per GPU, fft 2048, 4 distinct captures (noise + one cell) tiled" % caps},
           "captures_per_s": ctx.world * caps * steps / dt, "results_correct": bool(ok),
           "roofline": {"kernel": "srsran_hip_cellsearch_run (pss correlation + peak/PSR + SSS kernels; the correlation dominates)", "bound": "hbm",
                        "achieved": caps * cap_bytes / t_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": caps * cap_bytes / t_k / 1e9 / HBM_PEAK_GBS,
                        "traffic": tr, "traffic_source": src, "avg_launch_ms": t_k * 1e3, "algorithmic_bytes_per_launch": caps * cap_bytes,
                        "note": "algorithmic = every capture read once (2,457,600 B); events bracket the whole call"}}
    if want_cpu:
        import oracle_api as O

        x = d_caps[1].cpu().numpy().view(np.complex64).reshape(-1)
        O.pss_find_fft(x[:4096], N, 0)
        t0 = time.perf_counter()
        res = [O.pss_find_fft(x, N, n2) for n2 in range(3)]
        tc = time.perf_counter() - t0
        best = max(range(3), key=lambda n2: res[n2][1])
        par = all(res[n2][0] == got[3 + n2].peak_pos and abs(res[n2][1] - got[3 + n2].peak_value) <= 1e-3 * res[n2][1] for n2 in range(3))
        out["cpu_baseline"] = {"value": frame / tc / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                               "sample": "one capture, 3 N_id_2 hypotheses: srsran_pss_find_pss restated as ONE FFT convolution of length 309,248 per "
                                         "hypothesis (pss.c:446-534, convolution.c:113-120) on scipy.fft (pocketfft, complex64, one thread); the reference "
                                         "itself needs FFTW (absent)",
                               "parity_vs_gpu": "same peak positions, peak values within 1e-3" if par and best == 250 % 3 else "MISMATCH",
                               "reference_survey_time": {"value": 307200 / 31e-3 / 1e6, "unit": "Msamples/s",
                                                         "what": "reference srsran_pss_find_pss, 10.3 ms per N_id_2 (31 ms for three) on one core of an 8-vCPU "
                                                                 "Xeon @2.1 GHz with MKL's FFTW wrapper, measured at survey time (BASELINE.md par. 2), NOT on this box"}}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ------------------------------------------------------------------------------------------------ configs[3]: multi-UE uplink

def leg_uplink(ctx, steps=3, warmup=1, want_cpu=True, ues=64, sf=184, snr=19.0, iters=8, host_fed=True):
    """64 independent 20 MHz UEs per GPU x `sf` subframes each, every stage of the PUSCH receive path that is on the hot path:
    OFDM demodulation -> single-tap equaliser -> SC-FDMA transform de-precoding (1200-point IDFT) -> 64-QAM soft demodulation +
    descrambling -> rate de-matching + turbo decoding with CRC early stop + transport-block CRC.  The signal comes from the library's
    own transmit side; channel estimation (out of scope) is replaced by the known flat channel.
    sf = 184: 64 x 184 x 11 code blocks are eight rounds of the 2048 waves the early-stop decoder keeps resident.  The end of a launch runs
    on a part of the chip (waves take 2 ... 8 half iterations, CUs differ in speed): 46 subframes (two rounds) 44.8 Gbit/s, 92 50.6, 184 53.6
    on one box (tools/measure/uplink_sf.py)."""
    import torch

    import srslte_amd as S
    from srslte_amd import capi

    lib, dev, st = S.lib(), ctx.dev, ctx.stream
    nprb, nsc, mod, Qm = 100, 1200, 3, 6
    data_sym = [0, 1, 2, 4, 5, 6, 7, 8, 9, 11, 12, 13]  # PUSCH symbols of a subframe (3 and 10 carry the DMRS)
    n_re = len(data_sym) * nsc
    G = n_re * Qm
    tbs = 63776  # 11 code blocks of 5824 bits, no filler bits
    seg = capi.Cbsegm()
    assert lib.srsran_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0
    ncb = seg.C
    n_tb = ues * sf
    pool_n = 8
    rng = np.random.default_rng(100 + ctx.rank)
    # ---- transmit side (not timed): pool of transport blocks
    enc = C.c_void_p()
    capi.check(lib.srsran_hip_sch_enc_create(C.byref(enc)), "enc")
    payload = rng.integers(0, 256, (pool_n, tbs // 8)).astype(np.uint8)
    d_pay = torch.from_numpy(payload).to(dev)
    d_eb = torch.zeros((pool_n, G // 8), dtype=torch.uint8, device=dev)
    txd = (capi.HipTb * pool_n)(*[capi.HipTb(tbs, Qm, 0, G, i * G, i * (tbs // 8), 0) for i in range(pool_n)])
    capi.check(lib.srsran_hip_sch_encode(enc, d_pay.data_ptr(), txd, pool_n, d_eb.data_ptr(), st), "sch_encode")
    torch.cuda.synchronize()
    e = np.unpackbits(d_eb.cpu().numpy(), axis=1)
    seeds = [lib.srsran_hip_sequence_pusch_seed(0x200 + i, 2 * (i % 10), 42) for i in range(pool_n)]
    x = np.stack([_qam64(e[i] ^ _scrambling_bits(lib, seeds[i], G)) for i in range(pool_n)])
    d_x = torch.from_numpy(x.view(np.float32)).to(dev)  # [pool][n_re][2]
    fwd, inv = C.c_void_p(), C.c_void_p()
    capi.check(lib.srsran_hip_dft_batch_create(C.byref(fwd), nsc, capi.DFT_FORWARD, False, False, True), "dft fwd")  # dft_precoding.c: normalised
    capi.check(lib.srsran_hip_dft_batch_create(C.byref(inv), nsc, capi.DFT_BACKWARD, False, False, True), "dft inv")
    d_z = torch.zeros_like(d_x)
    capi.check(lib.srsran_hip_dft_batch_run(fwd, d_x.data_ptr(), d_z.data_ptr(), pool_n * len(data_sym), st), "precode")
    grid = torch.zeros((pool_n, 14, nsc, 2), dtype=torch.float32, device=dev)
    grid[:, data_sym] = d_z.view(pool_n, len(data_sym), nsc, 2)
    grid[:, [3, 10], :, 0] = 1.0  # placeholder reference symbols
    otx, orx = S.OfdmBatch(nprb, tx=True, normalize=True), S.OfdmBatch(nprb, normalize=True)
    d_time_pool = torch.zeros((pool_n, otx.sf_sz, 2), dtype=torch.float32, device=dev)
    otx.run(grid.data_ptr(), d_time_pool.data_ptr(), pool_n, st)
    d_g = torch.zeros((pool_n, 14, nsc, 2), dtype=torch.float32, device=dev)
    orx.run(d_time_pool.data_ptr(), d_g.data_ptr(), pool_n, st)
    torch.cuda.synchronize()
    gain = complex(float(d_g[:, [3, 10], :, 0].mean()), float(d_g[:, [3, 10], :, 1].mean()))  # flat channel = gain of modulator + demodulator
    reps = (n_tb + pool_n - 1) // pool_n
    d_time = d_time_pool.repeat(reps, 1, 1)[:n_tb].contiguous()
    sig = float(d_time.pow(2).sum(-1).mean().sqrt())
    sigma = sig * 10 ** (-snr / 20) / np.sqrt(2)
    d_time += sigma * torch.randn_like(d_time)
    # ---- receive side buffers
    d_grid = torch.zeros((n_tb, 14, nsc, 2), dtype=torch.float32, device=dev)
    d_h = torch.zeros((n_tb * n_re, 2), dtype=torch.float32, device=dev)
    d_h[:, 0], d_h[:, 1] = gain.real, gain.imag
    d_eq = torch.zeros((n_tb * n_re, 2), dtype=torch.float32, device=dev)
    d_sym = torch.zeros_like(d_eq)
    d_llr = torch.zeros((n_tb, G), dtype=torch.int16, device=dev)
    dlen = tbs // 8 + 8
    d_out = torch.zeros((n_tb, dlen), dtype=torch.uint8, device=dev)
    d_soft = torch.zeros((n_tb * ncb, capi.SOFTBUFFER_CB_SIZE), dtype=torch.int16, device=dev)
    flags = np.zeros(n_tb * ncb, np.uint8)
    res = (capi.HipTbResult * n_tb)()
    jobs = (capi.HipDemodJob * n_tb)(*[capi.HipDemodJob(mod, n_re, i * n_re, i * G, seeds[i % pool_n], 1) for i in range(n_tb)])
    rxd = (capi.HipTb * n_tb)(*[capi.HipTb(tbs, Qm, 0x100, G, i * G, i * dlen, i * ncb) for i in range(n_tb)])
    dem, sch = C.c_void_p(), C.c_void_p()
    capi.check(lib.srsran_hip_demod_create(C.byref(dem)), "demod")
    capi.check(lib.srsran_hip_sch_create(C.byref(sch)), "sch")
    idx = torch.tensor(data_sym, device=dev)
    noise_est = 0.0  # known channel, zero-forcing (pusch.c passes the estimator's figure)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    parts = []

    def step(i):
        flags[:] = 0
        # first transmission: SRSRAN_HIP_TB_NEW_DATA (0x100 in rv) stands for srsran_softbuffer_rx_reset
        ev[0].record()
        orx.run(d_time.data_ptr(), d_grid.data_ptr(), n_tb, st)
        ev[1].record()
        y = d_grid.index_select(1, idx).contiguous()  # the PUSCH symbols of every subframe (plumbing: a strided copy)
        capi.check(lib.srsran_hip_predecoding_single(y.data_ptr(), d_h.data_ptr(), d_eq.data_ptr(), None, n_tb * n_re, 1.0, noise_est, st), "eq")
        ev[2].record()
        capi.check(lib.srsran_hip_dft_batch_run(inv, d_eq.data_ptr(), d_sym.data_ptr(), n_tb * len(data_sym), st), "deprecode")
        ev[3].record()
        capi.check(lib.srsran_hip_demod_run(dem, d_sym.data_ptr(), d_llr.data_ptr(), capi.LLR_SHORT, jobs, n_tb, st), "demod")
        ev[4].record()
        capi.check(lib.srsran_hip_sch_decode(sch, d_llr.data_ptr(), rxd, n_tb, iters, d_soft.data_ptr(), flags.ctypes.data, d_out.data_ptr(), res, st),
                   "decode")
        ev[5].record()
        if i is not None:
            torch.cuda.synchronize()
            parts.append([ev[j].elapsed_time(ev[j + 1]) for j in range(5)])

    dt = _timed(ctx, torch, step, steps, warmup)
    (dt,) = ctx.max_over_ranks([dt])
    if ctx.rank != 0:
        return None
    pm = np.mean(np.array(parts), axis=0)
    ok = sum(1 for r in res if r.crc_ok == 0)
    got = d_out[:pool_n].cpu().numpy()
    good = all(np.array_equal(got[i][:tbs // 8], payload[i]) for i in range(pool_n) if res[i].crc_ok == 0)
    unit_bytes = otx.sf_sz * 8 + tbs // 8  # time samples of one UE-subframe in, payload bytes out
    t_step = dt / steps
    tr_u, src_u = traffic_of(load_traffic(), "uplink_chain", n_tb, "ue_subframes_per_step")
    out = {"metric": "multi-UE LTE uplink, PUSCH receive path from time samples to transport blocks, Mbit/s of TBS (all GPUs)",
           "value": ctx.world * n_tb * tbs * steps / dt / 1e6, "unit": "Mbit/s", "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
           "ms_per_step": t_step * 1e3, "timing": timing_fields(), "dtype": "f32 / int16", "scaling": "weak",
           "config": {"workload": "BASELINE configs[3], per-GPU shard: %d UEs x %d subframes, 20 MHz (100 PRB, symbol size N=%d: the reference's default "
                                  "for 100 PRB, phy_common.c:366-378), 64-QAM, TBS %d (%d code blocks of 5824), "
                                  "Es/N0 %.1f dB, at most %d half iterations with CRC early stop; %d distinct transport blocks (device transmit side + AWGN) tiled"
                                  % (ues, sf, otx.sf_sz // 15, tbs, ncb, snr, iters, pool_n)},
           "subframes_per_s": ctx.world * n_tb * steps / dt, "tb_crc_ok": [ok, n_tb], "payload_matches_on_ok_blocks": bool(good),
           "bler_gpu": 1.0 - ok / float(n_tb),
           "avg_half_iterations": float(np.mean([r.avg_iterations for r in res])),
           "stage_ms": {"ofdm_rx": float(pm[0]), "gather+equaliser": float(pm[1]), "transform_deprecoding": float(pm[2]),
                        "demod_descramble": float(pm[3]), "dematch_turbo_crc": float(pm[4])},
           "roofline": {"kernel": "whole chain (dominant: tdec_win_kernel<8, Ar16, true> inside dematch_turbo_crc)", "bound": "hbm",
                        "achieved": n_tb * unit_bytes / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": n_tb * unit_bytes / t_step / 1e9 / HBM_PEAK_GBS, "traffic": tr_u, "traffic_source": src_u, "avg_launch_ms": t_step * 1e3,
                        "algorithmic_bytes_per_launch": n_tb * unit_bytes,
                        "note": "algorithmic = time samples in (%d B: 15 x N=%d complex64) + payload out (%d B) per UE-subframe; wall time of the step, host work included"
                                % (otx.sf_sz * 8, otx.sf_sz // 15, tbs // 8)}}
    if host_fed and ctx.world == 1:
        out["host_fed"] = _uplink_host_fed(torch, S, capi, lib, dev, d_time, n_tb, otx.sf_sz, nprb, nsc, data_sym, n_re, G, tbs, ncb, mod, Qm, seeds, pool_n,
                                           gain, iters, payload, workers=3)
        out["tti"] = _uplink_host_fed(torch, S, capi, lib, dev, d_time, n_tb, otx.sf_sz, nprb, nsc, data_sym, n_re, G, tbs, ncb, mod, Qm, seeds, pool_n,
                                      gain, iters, payload, workers=3, chunks=96, per=ues, tti=True)
    if want_cpu:
        import oracle_api as O

        # one subframe through the CPU restatement of every stage: float stages against the device within 1e-4, integer stages bit for bit
        t0 = time.perf_counter()
        tsamp = d_time[0].cpu().numpy().view(np.complex64).reshape(-1)
        cfg = O.ofdm_cfg(nprb, normalize=1)
        g0 = O.ofdm_rx_fft(cfg, tsamp[None])[0].reshape(14, nsc)[data_sym].astype(np.complex128) / gain
        import scipy.fft as F

        z0 = (F.ifft(g0, axis=1, workers=1) * np.sqrt(nsc)).reshape(-1)
        s_dev = d_sym[:n_re].cpu().numpy().view(np.complex64).reshape(-1)
        par = np.abs(s_dev - z0).max() <= 1e-4 * max(1.0, float(np.abs(z0).max()))
        llr = O.sequence_apply(O.demod_soft(mod, s_dev, "s"), seeds[0])
        par = par and np.array_equal(d_llr[0].cpu().numpy(), llr)
        soft, crc = np.zeros((ncb, capi.SOFTBUFFER_CB_SIZE), np.int16), np.zeros(ncb, np.uint8)
        ret, data, avg = O.sch_decode_tb(tbs, Qm, 0, llr, soft, crc, iters)
        tc = time.perf_counter() - t0
        par = par and ret == res[0].crc_ok and abs(avg - res[0].avg_iterations) < 1e-6 and np.array_equal(data[:tbs // 8], got[0][:tbs // 8])
        port = {"value": tbs / tc / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "port",
                "sample": "one UE-subframe through the restated chain (scipy.fft OFDM + IDFT, C restatement of demodulator, de-matcher, "
                          "scalar-C window turbo decoder, CRCs), single thread",
                "parity_vs_gpu": "1e-4 on the de-precoded symbols, identical LLRs / verdict / iterations / bytes" if par else "MISMATCH"}
        out["cpu_baseline"] = port
        if O.have_ref() and _host_has_avx2():
            # the reference's OWN objects for everything that is not an FFT (oracle/_ref: demod_soft.c, sequence.c, rm_turbo.c, turbodecoder*.c,
            # crc.c, driven in the order of pusch.c:419-443 / sch.c:389-466); the two FFT stages by the scipy port (the reference's need FFTW)
            import ctypes as CC

            ref = CC.CDLL(O.REF_LIB)
            chain = O.RefSchChain(False, iters)
            n_s = 0
            llr_r = O.aligned_empty(G, np.int16)
            sym = O.aligned_empty(2 * n_re, np.float32)
            softr, same = chain.new_softbuffer(ncb), True
            ref_fail, ref_seen, gpu_fail_same = 0, set(), 0
            t0 = time.perf_counter()
            while n_s < 4 or time.perf_counter() - t0 < 3.0:
                i = n_s % min(n_tb, 64)
                ts_i = d_time[i].cpu().numpy().view(np.complex64).reshape(-1)
                gi = O.ofdm_rx_fft(cfg, ts_i[None])[0].reshape(14, nsc)[data_sym].astype(np.complex128) / gain
                sym[:] = (F.ifft(gi, axis=1, workers=1) * np.sqrt(nsc)).reshape(-1).astype(np.complex64).view(np.float32)
                assert ref.srsran_demod_soft_demodulate_s(mod, O.P(sym), O.P(llr_r), n_re) == 0
                ref.srsran_sequence_apply_s(O.P(llr_r), O.P(llr_r), CC.c_uint32(G), CC.c_uint32(seeds[i % pool_n]))
                softr[:] = 0
                crc_r = np.zeros(ncb, np.uint8)
                okr, data_r, avg_r = chain.decode_tb(tbs, Qm, 0, llr_r, softr, crc_r)
                if i == 0:
                    same = (0 if okr else -1) == res[0].crc_ok and abs(avg_r - res[0].avg_iterations) < 1e-6 and np.array_equal(data_r[:tbs // 8], got[0][:tbs // 8])
                if i not in ref_seen:  # block error rate on the distinct subframes the reference got through
                    ref_seen.add(i)
                    ref_fail += 0 if okr else 1
                    gpu_fail_same += 0 if res[i].crc_ok == 0 else 1
                n_s += 1
            tr = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n_s * tbs / tr / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "reference",
                                   "sample": "%d UE-subframes in %.1f s on one thread: the reference's own demod_soft / sequence / rm_turbo / turbodecoder (AUTO -> avx16 "
                                             "window) / crc objects (oracle/_ref) in the order of pusch.c:419-443 and sch.c:389-466; OFDM rx and the 1200-point IDFT by "
                                             "the scipy.fft port (the reference's dft_fftw.c needs FFTW, absent)" % (n_s, tr),
                                   "parity_vs_gpu": "identical verdict / iterations / bytes on the compared subframe" if same else "MISMATCH",
                                   "port_scalar_c": port}
            out["bler_reference"] = {"value": ref_fail / float(len(ref_seen)), "blocks": len(ref_seen), "gpu_on_the_same_blocks": gpu_fail_same / float(len(ref_seen))}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    lib.srsran_hip_sch_free(sch)
    return out


def leg_uplink_waterfall(ctx, steps=3, warmup=1, want_cpu=False):
    """the same chain at a waterfall operating point (Es/N0 16.4 dB: about 4 half iterations on average, some transport blocks fail) --
    the 19 dB point of `uplink` is the cheap regime (1.8 half iterations, every block decodes).  No CPU leg: `uplink` carries it."""
    out = leg_uplink(ctx, steps=steps, warmup=warmup, want_cpu=False, snr=16.4, host_fed=False)
    if out is not None:
        out["metric"] += " -- waterfall operating point"
    return out


# ------------------------------------------------------------------------------------------------ configs[1] through the 8-bit API

def leg_turbo8(ctx, steps=3, warmup=1, want_cpu=True, n_cb=131040, K=6144, nit=8):
    """BASELINE configs[1]'s decoder through the 8-bit API -- srsran_tdec_run_all_8bit is what srsenb and srsue run (cc_worker.cc sets
    llr_is_8bit in both): int8 LLRs, AUTO -> the 32-sub-block avx8 window decoder for K = 6144, 8 half iterations, natural-order input.
    CPU beside it: the reference's srsran_tdec_run_all_8bit (oracle/_ref) on one core, which is also the parity check."""
    import torch

    import srslte_amd as S
    from srslte_amd import capi
    import oracle_api as O

    lib, dev, st = S.lib(), ctx.dev, ctx.stream
    pool_n = 64
    _, hi = O.turbo_llrs_8bit(K, pool_n // 2, 3.0, seed=21 + ctx.rank)
    _, lo = O.turbo_llrs_8bit(K, pool_n // 2, -1.0, seed=22 + ctx.rank)
    pool = np.concatenate([hi, lo])
    reps = (n_cb + pool_n - 1) // pool_n
    d_llr = torch.from_numpy(pool).to(dev).repeat(reps, 1)[:n_cb].contiguous()
    d_bits = torch.zeros((n_cb, K // 8), dtype=torch.uint8, device=dev)
    dec = S.TdecBatch(K, n_cb, capi.TDEC_AUTO, llr8=True)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]

    def step(i):
        if i is not None:
            evs[i][0].record()
        capi.check(lib.srsran_hip_tdec_batch_run_8bit(dec._h, d_llr.data_ptr(), 3 * K + 12, d_bits.data_ptr(), K // 8, n_cb, nit, 0, st), "run8")
        if i is not None:
            evs[i][1].record()

    dt = _timed(ctx, torch, step, steps, warmup)
    t_k = sum(e[0].elapsed_time(e[1]) for e in evs) / steps * 1e-3
    dt, t_k = ctx.max_over_ranks([dt, t_k])
    if ctx.rank != 0:
        return None
    unit = 3 * K + 12 + K // 8  # SURVEY 8(d), int8 API: 19,212 B per block
    tr, src = traffic_of(load_traffic(), "tdec_win_kernel_8bit", n_cb, "code_blocks_per_launch")
    out = {"metric": "turbo decoded Mbit/s through the 8-bit API (LTE 20 MHz, K=6144, 8 half iterations; what srsenb / srsue run)",
           "value": ctx.world * n_cb * K * steps / dt / 1e6, "unit": "Mbit/s", "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "timing": timing_fields(), "dtype": "int8", "scaling": "weak",
           "config": {"workload": "srsran_hip_tdec_batch_run_8bit: %d blocks K=%d, nof_iterations=%d, AUTO -> avx8 window (32 sub-blocks); %d distinct noisy "
                                  "code words (half at Es/N0 3 dB, half at -1 dB) tiled %dx, per GPU" % (n_cb, K, nit, pool_n, reps)},
           "roofline": {"kernel": "tdec_win_kernel<32, Ar8, false>", "bound": "hbm", "achieved": n_cb * unit / t_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": n_cb * unit / t_k / 1e9 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src, "avg_launch_ms": t_k * 1e3,
                        "algorithmic_bytes_per_launch": n_cb * unit, "note": "VALU-issue bound (per-step max-normalisation), DESIGN.md par. 3.2"}}
    if want_cpu and O.have_ref() and _host_has_avx2():
        ref = C.CDLL(O.REF_LIB)
        h = C.create_string_buffer(64 * 1024)
        assert ref.srsran_tdec_init(h, K) == 0
        ref.srsran_tdec_force_not_sb(h)
        n = 48
        outs = np.zeros((n, K // 8), np.uint8)
        t0 = time.perf_counter()
        for i in range(n):
            assert ref.srsran_tdec_run_all_8bit(h, O.P(pool[i].copy()), O.P(outs[i]), nit, K) == 0
        again, scratch = 0, np.zeros(K // 8, np.uint8)
        while time.perf_counter() - t0 < 2.0:
            ref.srsran_tdec_run_all_8bit(h, O.P(pool[again % n].copy()), O.P(scratch), nit, K)
            again += 1
        tc = time.perf_counter() - t0
        ref.srsran_tdec_free(h)
        got = d_bits[:n].cpu().numpy()
        out["cpu_baseline"] = {"value": (n + again) * K / tc / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "reference",
                               "sample": "%d code blocks (%d distinct, compared with the device), reference srsran_tdec_run_all_8bit AUTO -> avx8 window "
                                         "(oracle/_ref), %.1f s on a single thread" % (n + again, n, tc),
                               "parity_vs_gpu": "bit-exact" if np.array_equal(outs, got) else "MISMATCH"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    del dec
    return out


# ------------------------------------------------------------------------------------------------ host-fed operation (PCIe inside the timed region)

def host_fed_turbo(S, capi, torch, dev, d_llr, in_stride, K, nit, llr8, chunk=8190, n_chunks=8, n_streams=3):
    """headline decoder fed from pinned host memory: `n_chunks` chunks of `chunk` code blocks, two streams with a device buffer each
    (H2D -> srsran_hip_tdec_batch_run{,_8bit} -> D2H per chunk, chunks take the streams in turn), so uploads overlap decodes.  Returns the rate, the one-stream
    (no overlap) rate of round 2 for comparison, the H2D rate of the same bytes alone and min(kernel rate, PCIe rate)."""
    import oracle_api as O

    lib = S.lib()
    chunk, n_streams = int(os.environ.get('HOSTFED_CHUNK', chunk)), int(os.environ.get('HOSTFED_STREAMS', n_streams))  # (tools/measure/hostfed.py sweeps them)
    n = chunk * n_chunks
    if llr8:
        _, pool = O.turbo_llrs_8bit(K, 64, 1.0, seed=5)
        h_in = torch.from_numpy(np.tile(pool, ((n + 63) // 64, 1))[:n].copy()).pin_memory()
        tdt = torch.int8
    else:
        h_in = torch.empty((n, in_stride), dtype=torch.int16).pin_memory()
        reps = (n + d_llr.shape[0] - 1) // d_llr.shape[0]
        h_in.copy_(d_llr.repeat(reps, 1)[:n] if reps > 1 else d_llr[:n])
        tdt = torch.int16
    esz = 1 if llr8 else 2
    h_out = torch.empty((n, K // 8), dtype=torch.uint8).pin_memory()
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    d_in = [torch.empty((chunk, in_stride), dtype=tdt, device=dev) for _ in range(n_streams)]
    d_out = [torch.empty((chunk, K // 8), dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    decs = [S.TdecBatch(K, chunk, capi.TDEC_AUTO, llr8=llr8) for _ in range(n_streams)]

    def decode(i, c):
        st = streams[i].cuda_stream
        if llr8:
            capi.check(lib.srsran_hip_tdec_batch_run_8bit(decs[i]._h, d_in[i].data_ptr(), in_stride, d_out[i].data_ptr(), K // 8, chunk, nit, 0, st), "run8")
        else:
            decs[i].run(d_in[i], in_stride, d_out[i], K // 8, chunk, nit, 0, st)

    def run(n_streams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(n_chunks):
            i = c % n_streams
            with torch.cuda.stream(streams[i]):
                d_in[i].copy_(h_in[c * chunk:(c + 1) * chunk], non_blocking=True)
                decode(i, c)
                h_out[c * chunk:(c + 1) * chunk].copy_(d_out[i], non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def h2d_only():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(n_chunks):
            i = c % n_streams
            with torch.cuda.stream(streams[i]):
                d_in[i].copy_(h_in[c * chunk:(c + 1) * chunk], non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def kernel_only():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(n_chunks):
            with torch.cuda.stream(streams[0]):
                decode(0, c)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(n_streams)
    t2 = min(run(n_streams) for _ in range(3))
    t1 = min(run(1) for _ in range(2))
    th = min(h2d_only() for _ in range(3))
    tk = min(kernel_only() for _ in range(2))
    bits = n * K
    bytes_in = n * in_stride * esz
    bound = bits / max(th, tk)
    out = {"value": bits / t2 / 1e6, "unit": "Mbit/s", "ms": t2 * 1e3,
           "what": "%d code blocks K=%d (%s LLRs) in %d chunks from pinned host memory: H2D of %.0f MB + decode (%d half iterations) + D2H of the bytes; %d streams, "
                   "a chunk's upload runs under the others' decodes (best of 3)" % (n, K, "int8" if llr8 else "int16", n_chunks, bytes_in / 1e6, nit, n_streams),
           "one_stream_no_overlap_mbit_per_s": bits / t1 / 1e6, "h2d_alone_gb_per_s": bytes_in / th / 1e9,
           "kernel_alone_mbit_per_s": bits / tk / 1e6, "bound_mbit_per_s": bound / 1e6, "bound": "pcie" if th > tk else "kernel",
           "frac_of_bound": (bits / t2) / bound}
    del decs
    return out


def _uplink_host_fed(torch, S, capi, lib, dev, d_time, n_tb, sf_sz, nprb, nsc, data_sym, n_re, G, tbs, ncb, mod, Qm, seeds, pool_n, gain, iters, payload,
                     workers=3, chunks=8, per=None, tti=False):
    """the uplink chain fed from HOST memory the way the reference's PHY is driven: `workers` threads (srsenb's nof_phy_threads, default 3), each with
    its own stream, handles and device buffers, take chunks of UE-subframes in turn: time samples pinned host -> HBM, the whole chain, payload bytes
    back to pinned host memory.  The transport-block call waits for its results (as decode_tb does), so overlap comes from the workers."""
    import threading

    per = per or n_tb // chunks
    n_use = per * chunks
    lat = []
    dlen = tbs // 8 + 8
    h_time = torch.empty((n_use, sf_sz, 2), dtype=torch.float32).pin_memory()
    h_time.copy_(d_time[:n_use])
    h_out = torch.empty((n_use, dlen), dtype=torch.uint8).pin_memory()
    idx = torch.tensor(data_sym, device=dev)

    class W:
        def __init__(self):
            self.st = torch.cuda.Stream(device=dev)
            self.orx = S.OfdmBatch(nprb, normalize=True)
            self.inv, self.dem, self.sch = C.c_void_p(), C.c_void_p(), C.c_void_p()
            capi.check(lib.srsran_hip_dft_batch_create(C.byref(self.inv), nsc, capi.DFT_BACKWARD, False, False, True), "dft inv")
            capi.check(lib.srsran_hip_demod_create(C.byref(self.dem)), "demod")
            capi.check(lib.srsran_hip_sch_create(C.byref(self.sch)), "sch")
            self.d_time = torch.empty((per, sf_sz, 2), dtype=torch.float32, device=dev)
            self.d_grid = torch.zeros((per, 14, nsc, 2), dtype=torch.float32, device=dev)
            self.d_h = torch.zeros((per * n_re, 2), dtype=torch.float32, device=dev)
            self.d_h[:, 0], self.d_h[:, 1] = gain.real, gain.imag
            self.d_eq = torch.zeros((per * n_re, 2), dtype=torch.float32, device=dev)
            self.d_sym = torch.zeros_like(self.d_eq)
            self.d_llr = torch.zeros((per, G), dtype=torch.int16, device=dev)
            self.d_out = torch.zeros((per, dlen), dtype=torch.uint8, device=dev)
            self.d_soft = torch.zeros((per * ncb, capi.SOFTBUFFER_CB_SIZE), dtype=torch.int16, device=dev)
            self.flags = np.zeros(per * ncb, np.uint8)
            self.res = (capi.HipTbResult * per)()
            self.rxd = (capi.HipTb * per)(*[capi.HipTb(tbs, Qm, 0x100, G, i * G, i * dlen, i * ncb) for i in range(per)])
            self.ok = 0

        def chunk(self, c):
            t_c = time.perf_counter()
            st = self.st.cuda_stream
            jobs = (capi.HipDemodJob * per)(*[capi.HipDemodJob(mod, n_re, i * n_re, i * G, seeds[(c * per + i) % pool_n], 1) for i in range(per)])
            with torch.cuda.stream(self.st):
                self.flags[:] = 0
                self.d_time.copy_(h_time[c * per:(c + 1) * per], non_blocking=True)
                self.orx.run(self.d_time.data_ptr(), self.d_grid.data_ptr(), per, st)
                y = self.d_grid.index_select(1, idx).contiguous()
                capi.check(lib.srsran_hip_predecoding_single(y.data_ptr(), self.d_h.data_ptr(), self.d_eq.data_ptr(), None, per * n_re, 1.0, 0.0, st), "eq")
                capi.check(lib.srsran_hip_dft_batch_run(self.inv, self.d_eq.data_ptr(), self.d_sym.data_ptr(), per * len(data_sym), st), "deprecode")
                capi.check(lib.srsran_hip_demod_run(self.dem, self.d_sym.data_ptr(), self.d_llr.data_ptr(), capi.LLR_SHORT, jobs, per, st), "demod")
                capi.check(lib.srsran_hip_sch_decode(self.sch, self.d_llr.data_ptr(), self.rxd, per, iters, self.d_soft.data_ptr(), self.flags.ctypes.data,
                                                     self.d_out.data_ptr(), self.res, st), "decode")
                h_out[c * per:(c + 1) * per].copy_(self.d_out, non_blocking=True)
                self.st.synchronize()
            lat.append(time.perf_counter() - t_c)
            self.ok += sum(1 for r in self.res if r.crc_ok == 0)

    ws = [W() for _ in range(workers)]

    def run(nw):
        errs = []

        def body(w, k):
            try:
                for c in range(k, chunks, nw):
                    w.chunk(c)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)

        for w in ws:
            w.ok = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=body, args=(ws[k], k)) for k in range(nw)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        if errs:
            raise errs[0]
        return time.perf_counter() - t0

    if tti:
        # one subframe's worth of UEs per call: latency of a call (host samples in -> payload on the host), 1 and `workers` threads
        res = {}
        for nw in (1, workers):
            run(nw)
            del lat[:]
            t = run(nw)
            v = sorted(lat)
            res["workers_%d" % nw] = {"p50_ms": 1e3 * v[len(v) // 2], "p99_ms": 1e3 * v[min(len(v) - 1, int(0.99 * len(v)))], "calls": len(v),
                                      "subframes_per_s": n_use / t, "mbit_per_s": n_use * tbs / t / 1e6}
        for w in ws:
            lib.srsran_hip_sch_free(w.sch)
        res["what"] = ("%d UE-subframes per call (one TTI of %d UEs): pinned host samples -> H2D -> OFDM -> equaliser -> IDFT -> demodulation -> transport blocks "
                       "(CRC early stop) -> payload D2H; latency of a call seen by the worker thread" % (per, per))
        return res
    run(workers)
    t3 = min(run(workers) for _ in range(2))
    ok3 = sum(w.ok for w in ws)
    t1 = run(1)
    # H2D of the same samples alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(chunks):
        ws[c % workers].d_time.copy_(h_time[c * per:(c + 1) * per], non_blocking=True)
    torch.cuda.synchronize()
    th = time.perf_counter() - t0
    good = bool(np.array_equal(h_out[0].numpy()[:tbs // 8], payload[0]))
    bytes_in = n_use * sf_sz * 8
    for w in ws:
        lib.srsran_hip_sch_free(w.sch)
    return {"value": n_use * tbs / t3 / 1e6, "unit": "Mbit/s", "ms": t3 * 1e3, "subframes_per_s": n_use / t3,
            "what": "%d UE-subframes in %d chunks from pinned host memory (%.0f MB of time samples in, payload bytes out), %d worker threads with a stream, handles "
                    "and buffers each (the reference's nof_phy_threads model); every chunk: H2D -> OFDM -> equaliser -> IDFT -> demodulation -> transport blocks -> D2H"
                    % (n_use, chunks, bytes_in / 1e6, workers),
            "one_worker_mbit_per_s": n_use * tbs / t1 / 1e6, "h2d_alone_gb_per_s": bytes_in / th / 1e9,
            "pcie_bound_mbit_per_s": n_use * tbs / th / 1e6, "tb_crc_ok": [ok3, n_use], "payload_matches": good}


# ------------------------------------------------------------------------------------------------ the reference's transport-block seams

def leg_seam(ctx, steps=3, warmup=1, want_cpu=True):
    """ONE transport block per call through the reference's own seams, on HOST buffers in the reference's structs -- what an unmodified
    srsran_pusch_decode / srsran_pdsch_nr_decode reaches (decode_tb_cb, sch.c:370; srsran_dlsch_nr_decode, sch_nr.c:724): p50 / p99 of 100 calls,
    host to host, beside the reference's own objects on one core for the same block (tools/seam_bench.py).  Latency, not throughput: `value` is
    the LTE 16-bit block's p50 in ms."""
    import oracle_api as O
    import seam_bench as SB
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    lte = SB.lte_points(lib, capi, O, 100, [8.0], with_ref=want_cpu)
    nr = SB.nr_points(lib, capi, O, 100, sigmas=(9.0,), with_ref=want_cpu)
    p16 = [p for p in lte["points"] if p["llr"] == "int16"][0]
    return {"metric": "one transport block per call through the reference's own seam, host to host (latency)", "value": p16["p50_ms"], "unit": "ms",
            "higher_is_better": False, "config": {"workload": lte["what"]}, "lte": lte["points"], "nr": nr}


# ------------------------------------------------------------------------------------------------ one device call per grant / per TTI

def leg_grant(ctx, steps=3, warmup=1, want_cpu=True):
    """The grant-level entry points (include/srsran_amd/phy_chan_abi.h) from HOST buffers, host to host, p50 / p99 of 100 calls: one full-band PUSCH grant
    per call; the 8 grants of a TTI one call each and all in ONE srsran_hip_pusch_decode_multi; the same for the PDSCH codewords of a TTI on the transmit
    side.  The signal is made by the library's own transmit side (srsran_hip_ulsch_encode -> srsran_hip_modulate_bytes with the PUSCH seed ->
    srsran_dft_precoding -> resource grid), every payload is checked.  Latency, not throughput: `value` is the full-band grant's p50 in ms.  The reference's
    own figure for the same grants (its pusch_test / pdsch_test stopwatch on one core) is in profiles/r04_ref_programs.json: 514 us (100 PRB), 32 us (12 PRB)."""
    import ctypes as C
    import time

    import numpy as np
    import srslte_amd as S
    from srslte_amd import capi

    lib = S.lib()
    SB, NPRB = 18600, 100
    pre = capi.DftPrecoding()
    capi.check(lib.srsran_dft_precoding_init(C.byref(pre), NPRB, True), "dft_precoding_init")

    def P(a):
        return a.ctypes.data_as(C.c_void_p)

    def tx_sb(n_cb):
        rows = [np.zeros(SB, np.uint8) for _ in range(n_cb)]
        return capi.SoftbufferTx(n_cb, SB, (C.c_void_p * n_cb)(*[r.ctypes.data for r in rows])), rows

    def rx_sb(n_cb):
        rows = [np.zeros(SB, np.int16) for _ in range(n_cb)]
        keep = [np.zeros(SB // 8, np.uint8) for _ in range(n_cb)]
        flags = np.zeros(n_cb, np.bool_)
        return capi.SoftbufferRx(n_cb, SB, (C.c_void_p * n_cb)(*[r.ctypes.data for r in rows]), (C.c_void_p * n_cb)(*[k.ctypes.data for k in keep]),
                                 flags.ctypes.data_as(C.POINTER(C.c_bool)), False), rows, keep, flags

    def make_ue(first_prb, L, mod, tbs, ident):
        nsymb, nsc, qm = 12, 12 * L, 2 * mod
        nof_re = nsymb * nsc
        payload = ((np.arange(tbs // 8) * 131 + ident * 17 + 5) & 0xff).astype(np.uint8)
        seed = lib.srsran_hip_sequence_pusch_seed(0x100 + ident, 4, 77)
        tb = capi.HipGrantTb(mod, tbs, 0, nof_re, seed, 10, 0, 1)
        sbt = tx_sb(13)
        q = np.zeros(nof_re * qm // 8 + 8, np.uint8)
        d, z = np.zeros(nof_re, np.complex64), np.zeros(nof_re, np.complex64)
        capi.check(lib.srsran_hip_ulsch_encode(C.byref(tb), nsymb, C.byref(sbt[0]), P(payload), P(q)), "ulsch_encode")
        assert lib.srsran_hip_modulate_bytes(mod, P(q), P(d), nof_re * qm, seed, 1, 1.0) == nof_re
        capi.check(lib.srsran_dft_precoding(C.byref(pre), P(d), P(z), L, nsymb), "dft_precoding")
        grid = np.zeros((14, 12 * NPRB), np.complex64)
        rows = [s for s in range(14) if s not in (3, 10)]
        grid[rows, 12 * first_prb:12 * first_prb + nsc] = z.reshape(nsymb, nsc)
        g = capi.HipPuschRx(tb, NPRB, 7, (C.c_uint32 * 2)(first_prb, first_prb), L, 0, 0.0, 0)
        return {"g": g, "grid": np.ascontiguousarray(grid.reshape(-1)), "ce": np.ones(14 * 12 * NPRB, np.complex64), "payload": payload, "out": np.zeros(tbs // 8 + 16, np.uint8),
                "rx": rx_sb(13), "tbs": tbs}

    def reset(u):
        for r in u["rx"][1]:
            r[:] = 0
        u["rx"][3][:] = False

    def pct(ts):
        ts = sorted(ts)
        return {"p50_us": ts[len(ts) // 2] * 1e6, "p99_us": ts[min(len(ts) - 1, int(len(ts) * 0.99))] * 1e6}

    calls = 100
    out = {}
    # ---- one full-band grant per call (12 code blocks of 6144)
    u = make_ue(0, 100, 3, 12 * (6144 - 24) - 24, 0)
    res = capi.HipGrantRes()
    ts, ok = [], True
    for i in range(calls + 10):
        reset(u)
        t0 = time.perf_counter()
        rc = lib.srsran_hip_pusch_decode(C.byref(u["g"]), P(u["grid"]), P(u["ce"]), C.byref(u["rx"][0]), P(u["out"]), C.byref(res))
        t1 = time.perf_counter()
        ok = ok and rc == 0 and res.crc_ok == 1 and np.array_equal(u["out"][:u["tbs"] // 8], u["payload"])
        if i >= 10:
            ts.append(t1 - t0)
    out["pusch_100prb"] = dict(pct(ts), ok=bool(ok), what="1 UE x 100 PRB, 64-QAM, TBS %d (12 code blocks of 6144), %.1f decoder half iterations per block" % (u["tbs"], res.avg_iterations_block))
    # ---- the 8 grants of a TTI (12 PRB each, 16-QAM, one code block of 5504): looped and in one call
    n = 8
    ues = [make_ue(12 * i, 12, 2, 5504 - 24, 10 + i) for i in range(n)]
    grants = (capi.HipPuschRx * n)(*[x["g"] for x in ues])
    gp = (C.c_void_p * n)(*[x["grid"].ctypes.data for x in ues])
    cp = (C.c_void_p * n)(*[x["ce"].ctypes.data for x in ues])
    sp = (C.POINTER(capi.SoftbufferRx) * n)(*[C.pointer(x["rx"][0]) for x in ues])
    dp = (C.c_void_p * n)(*[x["out"].ctypes.data for x in ues])
    rs = (capi.HipGrantRes * n)()
    tl, tm, ok = [], [], True
    for i in range(calls + 10):
        for x in ues:
            reset(x)
        t0 = time.perf_counter()
        for k, x in enumerate(ues):
            ok = ok and lib.srsran_hip_pusch_decode(C.byref(grants[k]), P(x["grid"]), P(x["ce"]), C.byref(x["rx"][0]), P(x["out"]), C.byref(rs[k])) == 0 and rs[k].crc_ok == 1
        t1 = time.perf_counter()
        for x in ues:
            ok = ok and np.array_equal(x["out"][:x["tbs"] // 8], x["payload"])
            x["out"][:] = 0
            reset(x)
        t2 = time.perf_counter()
        ok = ok and lib.srsran_hip_pusch_decode_multi(n, grants, gp, cp, sp, dp, rs) == 0
        t3 = time.perf_counter()
        for k, x in enumerate(ues):
            ok = ok and rs[k].crc_ok == 1 and np.array_equal(x["out"][:x["tbs"] // 8], x["payload"])
        if i >= 10:
            tl.append(t1 - t0)
            tm.append(t3 - t2)
    out["pusch_8x12prb_looped"] = dict(pct(tl), ok=bool(ok), what="8 UEs x 12 PRB, 16-QAM, one code block of 5504 each: one srsran_hip_pusch_decode per grant")
    out["pusch_8x12prb_one_call"] = dict(pct(tm), ok=bool(ok), what="the same 8 grants in ONE srsran_hip_pusch_decode_multi")
    # ---- transmit side: the PDSCH codewords of a TTI
    def make_cw(prb, mod, tbs, ident):
        nof_re = prb * 12 * 11
        return {"g": capi.HipPdschTx(capi.HipGrantTb(mod, tbs, 0, nof_re, 0x1234 + ident, 0, 0, 1), 1.0), "sb": tx_sb(13),
                "payload": ((np.arange(tbs // 8) * 37 + ident * 11 + 3) & 0xff).astype(np.uint8), "o1": np.zeros(nof_re, np.complex64), "o2": np.zeros(nof_re, np.complex64)}

    c100 = make_cw(100, 3, 75376, 0)
    ts, ok = [], True
    for i in range(calls + 10):
        t0 = time.perf_counter()
        ok = ok and lib.srsran_hip_pdsch_encode(C.byref(c100["g"]), C.byref(c100["sb"][0]), P(c100["payload"]), P(c100["o1"])) == 0
        t1 = time.perf_counter()
        if i >= 10:
            ts.append(t1 - t0)
    out["pdsch_encode_100prb"] = dict(pct(ts), ok=bool(ok), what="one codeword, 100 PRB, 64-QAM, TBS 75376 (13 code blocks)")
    cws = [make_cw(12, 2, 5736, 20 + i) for i in range(n)]
    gt = (capi.HipPdschTx * n)(*[x["g"] for x in cws])
    st = (C.POINTER(capi.SoftbufferTx) * n)(*[C.pointer(x["sb"][0]) for x in cws])
    pp = (C.c_void_p * n)(*[x["payload"].ctypes.data for x in cws])
    op = (C.c_void_p * n)(*[x["o2"].ctypes.data for x in cws])
    tl, tm, ok = [], [], True
    for i in range(calls + 10):
        t0 = time.perf_counter()
        for k, x in enumerate(cws):
            ok = ok and lib.srsran_hip_pdsch_encode(C.byref(gt[k]), C.byref(x["sb"][0]), P(x["payload"]), P(x["o1"])) == 0
        t1 = time.perf_counter()
        ok = ok and lib.srsran_hip_pdsch_encode_multi(n, gt, st, pp, op) == 0
        t2 = time.perf_counter()
        for x in cws:
            ok = ok and np.array_equal(x["o1"].view(np.uint32), x["o2"].view(np.uint32))
        if i >= 10:
            tl.append(t1 - t0)
            tm.append(t2 - t1)
    out["pdsch_encode_8x12prb_looped"] = dict(pct(tl), ok=bool(ok), what="8 codewords x 12 PRB, 16-QAM: one srsran_hip_pdsch_encode per codeword")
    out["pdsch_encode_8x12prb_one_call"] = dict(pct(tm), ok=bool(ok), what="the same 8 codewords in ONE srsran_hip_pdsch_encode_multi")
    lib.srsran_dft_precoding_free(C.byref(pre))
    return {"metric": "one full-band PUSCH grant per call through the grant-level entry point, host grids in, payload out (latency)", "value": out["pusch_100prb"]["p50_us"] / 1e3,
            "unit": "ms", "higher_is_better": False, "config": {"workload": out["pusch_100prb"]["what"]}, "calls": calls, "points": out,
            "all_payloads_ok": all(v["ok"] for v in out.values())}
