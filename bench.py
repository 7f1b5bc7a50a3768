#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X PHY DSP engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts N ranks ITSELF (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
environment) from a parent that never touches the GPU, waits for them and relays rank 0's JSON line; a failing rank fails
the run.  Under an external launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the
environment already carries the rank and the process is a worker.  `--gpus` must equal WORLD_SIZE.

Headline workload (BASELINE.json configs[1], LTE 20 MHz) -- per rank and per step, inputs resident in HBM:
    * srsran_ofdm_rx_sf on `--sf` subframes (2048-pt FFT, 100 PRB, normal CP, normalised)
    * srsran_tdec_run_all on 13 x `--sf` code blocks of K=6144, nof_iterations=8 (= 8 SISO runs = 4 full turbo
      iterations, the reference's unit; AUTO -> the 16-sub-block window decoder)
A step is one pass of that hot path over the batch.  `value` = decoded bits of all ranks / wall time of the timed region
(which contains the OFDM kernels too).  Units are independent: rank r owns the contiguous range
srslte_amd.sharding.shard_range(total, r, world) of a global unit list that grows with the world size (weak scaling);
the only collective of the job is one broadcast of the cell / decoder configuration at start-up (RCCL on GPUs).

The other single-GPU configurations of BASELINE.json ride in the same JSON line under `extra` (tools/bench_legs.py), each
with its own `roofline` and `cpu_baseline`: `ldpc` (configs[2]: NR 100 MHz, LDPC BG1 Z=384 at 20 iterations + OFDM N=4096),
`cellsearch` (configs[4]: PSS/SSS over 10 ms captures, 504 PCI hypotheses), `uplink` (configs[3]: the per-GPU shard of the
multi-UE PUSCH receive chain).  `--no-extras` runs the headline alone.

`roofline` describes the dominant kernel of the step (the turbo decoder), `roofline_ofdm` the OFDM demodulator (the HBM-bound
kernel of the path).  Kernel durations are measured live with HIP events on the launch stream.  `cpu_baseline` times the
reference's own turbo decoder (oracle/_ref, AVX2 host) or our scalar port (oracle/) on ONE host core over a bounded sample
of the same code blocks, and doubles as the in-bench parity check (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# (before torch initialises the HIP runtime: worker threads overlap only on separate hardware queues, srslte_amd/csrc/common.cpp)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
K_CB = 6144
NIT = 8
CB_PER_SF = 13  # 20 MHz, 64-QAM, MCS 28: TBS 75376 -> 13 code blocks (cbsegm.c:62-117)
N_FFT, N_PRB = 2048, 100
EXTRAS = ("ldpc", "cellsearch", "uplink", "uplink_waterfall", "turbo8", "seam", "grant")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    # 10,080 subframes = 131,040 code blocks = 16,380 waves of 8 blocks: eight full rounds of the 2048 waves the turbo kernel keeps resident.
    # The CUs do not finish their rounds together, so the end of a launch runs on a part of the chip: 4096 subframes (3.25 rounds) 30.2 Gbit/s,
    # 5040 (4 rounds) 33.5, 10,080 (8 rounds) 34.5, 20,160 (16 rounds) 35.2 on one box (DESIGN.md par. 3.2)
    ap.add_argument("--sf", type=int, default=10080, help="subframes per rank per step")
    ap.add_argument("--cpu-sample", type=int, default=48, help="code blocks decoded on the CPU for baseline + parity")
    ap.add_argument("--no-cpu", action="store_true", help="skip every cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="headline only (no extra.ldpc / cellsearch / uplink)")
    ap.add_argument("--only", default="", help="comma list of extras to run (default: all of %s)" % ",".join(EXTRAS))
    ap.add_argument("--extra-steps", type=int, default=8, help="timed steps of each extra leg (a cell-search step is 1.4 ms: three of them are mostly launch latency)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="CPU rehearsal of the N-rank launch path (gloo): launcher, rendezvous, config broadcast, sharding, "
                         "barrier + max-over-ranks timing; no kernels run and `value` is null")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher (GPU free)

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(a, argv):
    """parent of an N-rank run: imports neither torch nor the HIP library (a process that has touched the GPU must not
    spawn the ranks' interpreter by exec, and has no business holding a context on GPU 0 while rank 0 is timed)"""
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    failed = None
    out0 = []
    # rank 0's stdout is relayed line by line; a rank that dies takes the others down (they would wait in a barrier for ever)
    import threading

    def pump():
        for line in procs[0].stdout:
            out0.append(line)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    live = set(range(a.gpus))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0:
                    failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    t.join(timeout=10)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; run aborted\n" % failed)
        return failed[1] if failed[1] > 0 else 1
    return 0


# ------------------------------------------------------------------------------------------------ worker

class Ctx:
    """what a leg needs to know about the job: rank layout, device, stream, the two collectives the bench uses"""

    def __init__(self, rank, world, local, dev, cdev, dist, stream):
        self.rank, self.world, self.local, self.dev, self.cdev, self.dist, self.stream = rank, world, local, dev, cdev, dist, stream

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, values):
        from srslte_amd import sharding

        return [sharding.max_over_ranks(v, self.cdev) for v in values]


def host_has_avx2():
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2 " in f.read().replace("\n", " ")
    except OSError:
        return False


def cpu_baseline(llr_np, n_sample):
    """decode n_sample code blocks on ONE host core; returns (bytes_out, info dict)"""
    import ctypes as C

    import numpy as np
    import oracle_api as O

    n = min(n_sample, llr_np.shape[0])
    out = np.zeros((n, K_CB // 8), np.uint8)
    kind = "port"
    if O.have_ref() and host_has_avx2():
        kind = "reference"
        ref = C.CDLL(O.REF_LIB)
        h = C.create_string_buffer(64 * 1024)
        assert ref.srsran_tdec_init(h, K_CB) == 0
        ref.srsran_tdec_force_not_sb(h)
        t0 = time.perf_counter()
        for i in range(n):
            assert ref.srsran_tdec_run_all(h, O.P(llr_np[i]), O.P(out[i]), NIT, K_CB) == 0
        # ... and around the same blocks again until the sample is about two seconds of CPU work (n blocks are a few milliseconds)
        again, scratch = 0, np.zeros(K_CB // 8, np.uint8)
        while time.perf_counter() - t0 < 2.0:
            assert ref.srsran_tdec_run_all(h, O.P(llr_np[again % n]), O.P(scratch), NIT, K_CB) == 0
            again += 1
        dt = time.perf_counter() - t0
        ref.srsran_tdec_free(h)
        n_timed = n + again
    else:
        t0 = time.perf_counter()
        out = O.turbo_decode(llr_np[:n], NIT, K_CB)
        dt = time.perf_counter() - t0
        n_timed = n
    info = {"value": n_timed * K_CB / dt / 1e6, "unit": "Mbit/s", "cores": 1, "kind": kind,
            "sample": "%d code blocks K=%d (%d distinct, compared with the device), nof_iterations=%d, %.1f s on a single thread (%s)" %
                      (n_timed, K_CB, n, NIT, dt, "reference srsran_tdec_run_all AUTO->avx16 window, oracle/_ref" if kind == "reference"
                       else "scalar C restatement, oracle/")}
    return out, info


def ofdm_cpu_port(n_prb, n_fft, n_sf=64):
    """srsran_ofdm_rx_sf restated on scipy's pocketfft (complex64, one thread) -- the reference itself needs FFTW (absent);
    pinned to the C oracle by tests/test_oracle_golden.py::test_fft_ports_match_the_oracle"""
    import numpy as np
    import oracle_api as O

    cfg = O.ofdm_cfg(n_prb, n_fft, 0, 1)
    n, nsym, sf_sz, sf_re = O.ofdm_geometry(cfg)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((n_sf, sf_sz)) + 1j * rng.standard_normal((n_sf, sf_sz))).astype(np.complex64)
    O.ofdm_rx_fft(cfg, x[:2])
    t0 = time.perf_counter()
    O.ofdm_rx_fft(cfg, x)
    dt = time.perf_counter() - t0
    return {"value": n_sf * sf_sz / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d subframes N=%d %d PRB, scipy.fft (pocketfft) complex64, one thread; the reference's FFTW path cannot be "
                      "built here (survey-time figure with MKL's FFTW wrapper on another host: ~675 Msamples/s, BASELINE.md par. 2)"
                      % (n_sf, n_fft, n_prb)}


def make_turbo_pool(S, capi, torch, dev, stream, k_cb, pool_n, seed):
    """distinct noisy code words made by the library's own encoder (srsran_hip_tcod_encode_batch) + device AWGN:
    half at an error-free Es/N0 (3 dB), half in the waterfall (-1 dB); int16 LLR = round(100 y) as turbodecoder_test.c:254"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    msgs = torch.randint(0, 2, (pool_n, k_cb), generator=g, device=dev, dtype=torch.uint8)
    enc = torch.zeros((pool_n, 3 * k_cb + 12), dtype=torch.uint8, device=dev)
    capi.check(S.lib().srsran_hip_tcod_encode_batch(msgs.data_ptr(), k_cb, enc.data_ptr(), 3 * k_cb + 12, pool_n, k_cb, stream), "tcod_encode_batch")
    torch.cuda.synchronize()
    sigma = torch.full((pool_n, 1), 10 ** (-3.0 / 20), device=dev)
    sigma[pool_n // 2:] = 10 ** (1.0 / 20)
    y = 2.0 * enc.float() - 1.0 + sigma * torch.randn((pool_n, 3 * k_cb + 12), generator=g, device=dev)
    return torch.clamp(torch.round(100.0 * y), -32768, 32767).to(torch.int16), msgs


def plumbing_only(a, rank, world):
    """the N-rank path without a GPU: same launcher, rendezvous, broadcast, sharding and timing protocol, gloo backend"""
    import torch
    import torch.distributed as dist
    from srslte_amd import sharding

    if world > 1:  # the workers' own rendezvous (sharding.job_store: also under torch.distributed.run, whose agent owns MASTER_PORT)
        sharding.init_collectives(rank, world, torch.device("cpu"), prefer="gloo")
    cfg = sharding.broadcast_config({"n_prb": N_PRB, "n_fft": N_FFT, "k_cb": K_CB, "nit": NIT, "cb_per_sf": CB_PER_SF, "sf": a.sf} if rank == 0 else None)
    lo, hi = sharding.shard_range(cfg["sf"] * world, rank, world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))  # stands for the K steps: the slowest rank sets the time
    if world > 1:
        dist.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0)
    ranges = [None] * world
    if world > 1:
        dist.all_gather_object(ranges, (lo, hi))
    else:
        ranges = [(lo, hi)]
    if rank == 0:
        emit({"metric": "plumbing rehearsal (no kernels)", "value": None, "unit": "Mbit/s", "n_gpus": world, "steps": a.steps,
              "warmup": a.warmup, "plumbing_only": True, "config": cfg, "shards": ranges, "max_step_time_s": dt,
              "torch": torch.__version__})
    if world > 1:
        dist.destroy_process_group()


_RESULT_FD = None


def _quiet_stdout():
    """Only the JSON line may reach stdout: native libraries write there too (gloo: "[Gloo] Rank 0 is connected to 1 peer ranks ...").
    The process's stdout is kept aside for emit() and file descriptor 1 points at stderr from here on."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def worker(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    _quiet_stdout()
    if a.plumbing_only:
        return plumbing_only(a, rank, world)
    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the PHY engine has no CPU fallback")
    # rehearsal switch for a 1-GPU box: all ranks share GPU 0 and the (tiny) collectives run over gloo
    share_gpu = os.environ.get("SRSLTE_AMD_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible); set SRSLTE_AMD_BENCH_SHARE_GPU=1 to rehearse "
                         "on one GPU" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if share_gpu else dev  # where collective payloads live
    from srslte_amd import sharding

    # The job's only collectives -- one configuration broadcast, barriers, a max of a few timings -- go through RCCL, at world size 1
    # as well: a 1-GPU run executes the same bootstrap (librccl, communicator on this rank's device, first all-reduce) an 8-GPU run needs.
    # If RCCL cannot be brought up the ranks AGREE on gloo through the rendezvous store before anyone re-initialises
    # (sharding.init_collectives) -- the ranks exchange no data, so the measurement stands -- and the JSON line says so.
    backend, cdev = sharding.init_collectives(rank, world, dev, prefer="gloo" if share_gpu else "nccl")
    if share_gpu:
        backend = "gloo (ranks share one GPU)"
    seen = sharding.ranks_seen(local, dev, cdev)

    import srslte_amd as S
    from srslte_amd import capi

    S.capi.check(S.lib().srsran_hip_set_device(local), "set_device")
    stream = torch.cuda.current_stream().cuda_stream
    ctx = Ctx(rank, world, local, dev, cdev, dist, stream)

    # ---- the only collective of the job: rank 0 broadcasts the cell / decoder configuration
    cfg = sharding.broadcast_config({"n_prb": N_PRB, "n_fft": N_FFT, "k_cb": K_CB, "nit": NIT, "cb_per_sf": CB_PER_SF, "sf": a.sf}
                                    if rank == 0 else None, cdev)
    n_prb, n_fft, k_cb, nit, cb_per_sf = cfg["n_prb"], cfg["n_fft"], cfg["k_cb"], cfg["nit"], cfg["cb_per_sf"]
    # the job's unit list is `sf x world` subframes; this rank owns a contiguous range of it
    lo, hi = sharding.shard_range(cfg["sf"] * world, rank, world)
    n_sf = hi - lo
    n_cb = n_sf * cb_per_sf

    # ---- synthetic inputs, resident in HBM before the timed region
    pool_n = 64
    d_pool, d_msgs = make_turbo_pool(S, capi, torch, dev, stream, k_cb, pool_n, 1000 + rank)
    in_stride = 3 * k_cb + 12
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

    def step(ev=None):
        if ev:
            ev[0].record()
        ofdm.run(d_time, d_re, n_sf, stream)
        if ev:
            ev[1].record()
        tdec.run(d_llr, in_stride, d_bits, k_cb // 8, n_cb, nit, 0, stream)
        if ev:
            ev[2].record()

    import bench_legs as L

    # warm-up, barrier + synchronize, EXACTLY `steps` steps, synchronize + barrier (bench_legs._timed: the same protocol for every leg);
    # next to the wall time it leaves per-step end-event deltas, the device span and the cost of the closing barrier in `timing`
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    dt = L._timed(ctx, torch, lambda i: step(None if i is None else evs[i]), a.steps, a.warmup)
    timing = L.timing_fields()
    t_ofdm = sum(e[0].elapsed_time(e[1]) for e in evs) / a.steps * 1e-3
    t_tdec = sum(e[1].elapsed_time(e[2]) for e in evs) / a.steps * 1e-3
    dt, t_ofdm, t_tdec = ctx.max_over_ranks([dt, t_ofdm, t_tdec])

    res = None
    if rank == 0:
        total_bits = float(n_cb) * k_cb * a.steps * world
        value = total_bits / dt / 1e6
        # algorithmic bytes per unit: turbo in (3K+12)*2 + out K/8 ; OFDM 8*(15N + 14*12*PRB)
        cb_bytes = (3 * k_cb + 12) * 2 + k_cb // 8
        sf_bytes = 8 * (15 * n_fft + 14 * 12 * n_prb)
        r_t = n_cb * cb_bytes / t_tdec / 1e9
        r_o = n_sf * sf_bytes / t_ofdm / 1e9
        # HBM traffic per launch measured with rocprofv3 PMC on this very command (profiles/r0X_traffic.json: FETCH_SIZE
        # and WRITE_SIZE in separate passes, gfx950 read correction applied); scaled if the batch size was overridden
        import bench_legs as L

        tj = L.load_traffic()
        traffic_t, tnote = L.traffic_of(tj, "tdec_win_kernel", n_cb, "code_blocks_per_launch")
        traffic_o, _ = L.traffic_of(tj, "ofdm_kernel", n_sf, "subframes_per_launch")
        res = {
            "metric": "turbo decoded Mbit/s (LTE 20 MHz, K=6144, 8 half-iterations) incl. OFDM demod of the same subframes",
            "value": value, "unit": "Mbit/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "timing": timing, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic", "collective_backend": backend, "ranks_seen": seen,
            "config": {"workload": "LTE 20 MHz (BASELINE configs[1]): ofdm_rx_sf N=2048 100 PRB + tdec_run_all K=6144 nof_iterations=8, "
                                   "%d subframes + %d code blocks per GPU per step; the code blocks are %d distinct noisy code words "
                                   "(device encoder + AWGN: half at Es/N0 3 dB, half at -1 dB) tiled %dx -- a fixed-iteration decoder "
                                   "does the same work on every block" % (n_sf, n_cb, pool_n, reps),
                       "subframes_per_gpu": n_sf, "code_blocks_per_gpu": n_cb, "distinct_code_words": pool_n},
            "ofdm_msamples_per_s": n_sf * ofdm.sf_sz * world / t_ofdm / 1e6,
            "turbo_kernel_mbit_per_s": n_cb * k_cb * world / t_tdec / 1e6,
            "roofline": {"kernel": "tdec_win_kernel<8>", "bound": "hbm", "achieved": r_t, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": r_t / HBM_PEAK_GBS, "traffic": traffic_t,
                         "avg_launch_ms": t_tdec * 1e3, "algorithmic_bytes_per_launch": n_cb * cb_bytes,
                         "traffic_rate_gbs": (traffic_t / t_tdec / 1e9) if traffic_t else None, "traffic_source": tnote,
                         "note": "iterative decoder: 8 half iterations over a per-code-block workspace (LLRs, extrinsics, "
                                 "check-points); what of it streams through HBM is `traffic` (DESIGN.md par. 3.2)"},
            "roofline_ofdm": {"kernel": "ofdm_kernel<Plan<2048,...>,rx>", "bound": "hbm", "achieved": r_o,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_o / HBM_PEAK_GBS, "traffic": traffic_o,
                              "avg_launch_ms": t_ofdm * 1e3, "algorithmic_bytes_per_launch": n_sf * sf_bytes,
                              "traffic_rate_gbs": (traffic_o / t_ofdm / 1e9) if traffic_o else None},
        }
        if not a.no_cpu and world == 1:
            pool = d_pool.cpu().numpy()
            cpu_bits, info = cpu_baseline(pool, a.cpu_sample)
            gpu_bits = d_bits[:pool_n].cpu().numpy()[:cpu_bits.shape[0]]
            info["parity_vs_gpu"] = "bit-exact" if np.array_equal(cpu_bits, gpu_bits) else "MISMATCH"
            # block error rate of both decoders against the transmitted messages ("at matching BLER"): the 64 distinct code words of the step on the
            # device, the sampled ones on the reference -- half of the pool sits in the waterfall (Es/N0 -1 dB), so the rate is not trivially zero
            sent = np.packbits(d_msgs.cpu().numpy(), axis=1)
            wrong_gpu = (d_bits[:pool_n].cpu().numpy() != sent).any(axis=1)
            wrong_ref = (cpu_bits != sent[:cpu_bits.shape[0]]).any(axis=1)
            res["bler"] = {"gpu": float(wrong_gpu.mean()), "gpu_blocks": int(pool_n), "reference": float(wrong_ref.mean()), "reference_blocks": int(cpu_bits.shape[0]),
                           "gpu_on_reference_sample": float(wrong_gpu[:cpu_bits.shape[0]].mean()),
                           "same_blocks_wrong": bool(np.array_equal(wrong_gpu[:cpu_bits.shape[0]], wrong_ref)),
                           "note": "%d distinct code words, half at Es/N0 3 dB, half at -1 dB; %d half iterations, no early stop" % (pool_n, nit)}
            res["cpu_baseline"] = info
            res["speedup_vs_cpu_baseline"] = value / info["value"]
            if info["parity_vs_gpu"] != "bit-exact":
                res["error"] = "GPU hard decisions differ from the CPU decoder on the sampled code blocks"
            res["roofline_ofdm"]["cpu_port"] = ofdm_cpu_port(n_prb, n_fft)

    # ---- PCIe-inclusive rates (never `value`): inputs start in pinned HOST memory, results end there.  Chunks alternate between two streams
    # (each: H2D -> decode -> D2H), so the upload of one chunk runs under the decode of the other; the H2D rate of the same bytes alone and
    # the device-resident kernel rate bound it from above.
    if world == 1 and not a.no_extras:
        import bench_legs as L

        res["pcie_inclusive"] = L.host_fed_turbo(S, capi, torch, dev, d_llr, in_stride, k_cb, nit, llr8=False, chunk=16380)
        res["pcie_inclusive_8bit"] = L.host_fed_turbo(S, capi, torch, dev, None, 3 * k_cb + 12, k_cb, nit, llr8=True, chunk=8190)

    # ---- the other single-GPU configurations of BASELINE.json
    del tdec, d_llr, d_bits, d_time, d_re
    torch.cuda.empty_cache()
    if not a.no_extras:
        import bench_legs as L

        which = [w for w in (a.only.split(",") if a.only else EXTRAS) if w]
        extra = {}
        for name in which:
            if name not in EXTRAS:
                raise SystemExit("bench.py: unknown extra %r" % name)
            out = getattr(L, "leg_" + name)(ctx, steps=a.extra_steps, warmup=1, want_cpu=(not a.no_cpu and world == 1))
            if rank == 0:
                extra[name] = out
            torch.cuda.empty_cache()
        if rank == 0:
            res["extra"] = extra
    if rank == 0:
        emit(res)
    dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(a, argv))
    worker(a)


if __name__ == "__main__":
    main()
