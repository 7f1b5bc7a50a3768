#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X PHY DSP engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LTE 20 MHz -- per rank and per step, with inputs resident in HBM:
    * srsran_ofdm_rx_sf on `--sf` subframes (2048-pt FFT, 100 PRB, normal CP, normalised)
    * srsran_tdec_run_all on 13 x `--sf` code blocks of K=6144, nof_iterations=8 (= 8 SISO runs =
      4 full turbo iterations, the reference's unit; AUTO -> the 16-sub-block window decoder)
A step is one pass of that hot path over the batch.  `value` = decoded bits of all ranks / wall time of
the timed region (which contains the OFDM kernels too).  Ranks are independent (weak scaling); the
only collective is one RCCL broadcast of the cell/decoder configuration at start-up.

One JSON line is printed by rank 0.  `roofline` describes the dominant kernel of the step (the turbo
decoder); `roofline_ofdm` the OFDM demodulator (the HBM-bound kernel of the path).  Kernel durations
are measured live with HIP events on the launch stream.  `cpu_baseline` times the reference's own
turbo decoder (oracle/_ref, when present and the host has AVX2) or our scalar port (oracle/) on ONE
host core over a bounded sample of the same code blocks, and doubles as the in-bench parity check.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
K_CB = 6144
NIT = 8
CB_PER_SF = 13  # 20 MHz, 64-QAM, MCS 28: TBS 75376 -> 13 code blocks (cbsegm.c:62-117)
N_FFT, N_PRB = 2048, 100


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    # 5040 subframes = 65,520 code blocks = 8190 waves of 8 blocks: four full rounds of the 2048 waves the turbo kernel keeps resident
    # (4096 subframes = 3.25 rounds leave the chip three quarters empty for the last round: 30.2 instead of 33.0 Gbit/s)
    ap.add_argument("--sf", type=int, default=5040, help="subframes per rank per step")
    ap.add_argument("--cpu-sample", type=int, default=48, help="code blocks decoded on the CPU for baseline + parity")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def host_has_avx2():
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2 " in f.read().replace("\n", " ")
    except OSError:
        return False


def cpu_baseline(llr_np, n_sample):
    """decode n_sample code blocks on ONE host core; returns (bytes_out, info dict)"""
    import oracle_api as O

    n = min(n_sample, llr_np.shape[0])
    out = np.zeros((n, K_CB // 8), np.uint8)
    kind = "port"
    t0 = time.perf_counter()
    if O.have_ref() and host_has_avx2():
        kind = "reference"
        ref = C.CDLL(O.REF_LIB)
        h = C.create_string_buffer(64 * 1024)
        assert ref.srsran_tdec_init(h, K_CB) == 0
        ref.srsran_tdec_force_not_sb(h)
        t0 = time.perf_counter()
        for i in range(n):
            assert ref.srsran_tdec_run_all(h, O.P(llr_np[i]), O.P(out[i]), NIT, K_CB) == 0
        dt = time.perf_counter() - t0
        ref.srsran_tdec_free(h)
    else:
        t0 = time.perf_counter()
        out = O.turbo_decode(llr_np[:n], NIT, K_CB)
        dt = time.perf_counter() - t0
    info = {"value": n * K_CB / dt / 1e6, "unit": "Mbit/s", "cores": 1, "kind": kind,
            "sample": "%d code blocks K=%d, nof_iterations=%d, single thread (%s)" %
                      (n, K_CB, NIT, "reference srsran_tdec_run_all AUTO->avx16 window, oracle/_ref" if kind == "reference"
                       else "scalar C restatement, oracle/")}
    return out, info


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the PHY engine has no CPU fallback")
    # rehearsal switch for a 1-GPU box: all ranks share GPU 0 and the (tiny) collectives run over gloo
    share_gpu = os.environ.get("SRSLTE_AMD_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if share_gpu else dev  # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import srslte_amd as S
    from srslte_amd import capi
    import oracle_api as O

    S.capi.check(S.lib().srsran_hip_set_device(local), "set_device")

    # ---- the only collective of the job: rank 0 broadcasts the cell / decoder configuration
    cfg = torch.tensor([N_PRB, N_FFT, K_CB, NIT, CB_PER_SF, a.sf], dtype=torch.int32, device=cdev)
    if world > 1:
        dist.broadcast(cfg, src=0)
    n_prb, n_fft, k_cb, nit, cb_per_sf, n_sf = [int(v) for v in cfg.tolist()]
    n_cb = n_sf * cb_per_sf

    # ---- synthetic inputs, resident in HBM before the timed region
    # turbo: a pool of distinct noisy code words (half error-free Es/N0, half in the waterfall), tiled
    pool_n = 64
    _, llr_a = O.turbo_llrs(k_cb, pool_n // 2, 3.0, seed=1000 + rank)
    _, llr_b = O.turbo_llrs(k_cb, pool_n // 2, -1.0, seed=2000 + rank)
    pool = np.concatenate([llr_a, llr_b], axis=0)
    in_stride = 3 * k_cb + 12
    d_pool = torch.from_numpy(pool).to(dev)
    reps = (n_cb + pool_n - 1) // pool_n
    d_llr = d_pool.repeat(reps, 1)[:n_cb].contiguous()
    d_bits = torch.zeros((n_cb, k_cb // 8), dtype=torch.uint8, device=dev)
    # OFDM: unit-variance complex Gaussian time samples
    ofdm = S.OfdmBatch(n_prb, tx=False, symbol_sz=n_fft, normalize=True)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    d_time = torch.view_as_complex(torch.randn((n_sf, ofdm.sf_sz, 2), generator=g, device=dev, dtype=torch.float32) * 0.7071)
    d_re = torch.zeros((n_sf, ofdm.sf_re), dtype=torch.complex64, device=dev)
    tdec = S.TdecBatch(k_cb, n_cb, capi.TDEC_AUTO)

    stream = torch.cuda.current_stream().cuda_stream

    def step(ev=None):
        if ev:
            ev[0].record()
        ofdm.run(d_time, d_re, n_sf, stream)
        if ev:
            ev[1].record()
        tdec.run(d_llr, in_stride, d_bits, k_cb // 8, n_cb, nit, 0, stream)
        if ev:
            ev[2].record()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(evs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_ofdm = sum(e[0].elapsed_time(e[1]) for e in evs) / a.steps * 1e-3
    t_tdec = sum(e[1].elapsed_time(e[2]) for e in evs) / a.steps * 1e-3
    tt = torch.tensor([dt, t_ofdm, t_tdec], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, t_ofdm, t_tdec = [float(v) for v in tt.tolist()]

    if rank == 0:
        total_bits = float(n_cb) * k_cb * a.steps * world
        value = total_bits / dt / 1e6
        # algorithmic bytes per unit: turbo in (3K+12)*2 + out K/8 ; OFDM 8*(15N + 14*12*PRB)
        cb_bytes = (3 * k_cb + 12) * 2 + k_cb // 8
        sf_bytes = 8 * (15 * n_fft + 14 * 12 * n_prb)
        r_t = n_cb * cb_bytes / t_tdec / 1e9
        r_o = n_sf * sf_bytes / t_ofdm / 1e9
        # HBM traffic per launch measured with rocprofv3 PMC on this very command (profiles/r01_traffic.json: FETCH_SIZE
        # and WRITE_SIZE in separate passes, gfx950 read correction applied); scaled if the batch size was overridden
        traffic_t = traffic_o = None
        tnote = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            traffic_t = tj["tdec_win_kernel"]["traffic_bytes_per_launch"] * n_cb / float(tj["tdec_win_kernel"]["code_blocks_per_launch"])
            traffic_o = tj["ofdm_kernel"]["traffic_bytes_per_launch"] * n_sf / float(tj["ofdm_kernel"]["subframes_per_launch"])
            tnote = tj["source"]
        except Exception:
            pass
        res = {
            "metric": "turbo decoded Mbit/s (LTE 20 MHz, K=6144, 8 half-iterations) incl. OFDM demod of the same subframes",
            "value": value, "unit": "Mbit/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": "LTE 20 MHz: ofdm_rx_sf N=2048 100 PRB + tdec_run_all K=6144 nof_iterations=8, "
                                   "%d subframes + %d code blocks per GPU per step" % (n_sf, n_cb),
                       "subframes_per_gpu": n_sf, "code_blocks_per_gpu": n_cb},
            "ofdm_msamples_per_s": n_sf * ofdm.sf_sz * world / t_ofdm / 1e6,
            "turbo_kernel_mbit_per_s": n_cb * k_cb * world / t_tdec / 1e6,
            "roofline": {"kernel": "tdec_win_kernel<8>", "bound": "hbm", "achieved": r_t, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": r_t / HBM_PEAK_GBS, "traffic": traffic_t,
                         "avg_launch_ms": t_tdec * 1e3, "algorithmic_bytes_per_launch": n_cb * cb_bytes,
                         "traffic_rate_gbs": (traffic_t / t_tdec / 1e9) if traffic_t else None, "traffic_source": tnote,
                         "note": "iterative decoder: 8 half iterations stream a 74 KB-per-code-block workspace (LLRs, extrinsics, "
                                 "check-points) that cannot stay on chip for 16k blocks in flight; that workspace traffic, not the "
                                 "algorithmic input/output bytes, is what the kernel is bound by (DESIGN.md)"},
            "roofline_ofdm": {"kernel": "ofdm_kernel<Plan<2048,...>,rx>", "bound": "hbm", "achieved": r_o,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_o / HBM_PEAK_GBS, "traffic": traffic_o,
                              "avg_launch_ms": t_ofdm * 1e3, "algorithmic_bytes_per_launch": n_sf * sf_bytes},
        }
        if not a.no_cpu:
            cpu_bits, info = cpu_baseline(pool, a.cpu_sample)
            gpu_bits = d_bits[:pool_n].cpu().numpy()[:cpu_bits.shape[0]]
            info["parity_vs_gpu"] = "bit-exact" if np.array_equal(cpu_bits, gpu_bits) else "MISMATCH"
            res["cpu_baseline"] = info
            res["speedup_vs_cpu_baseline"] = value / world / info["value"]
            if info["parity_vs_gpu"] != "bit-exact":
                res["error"] = "GPU hard decisions differ from the CPU decoder on the sampled code blocks"
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
