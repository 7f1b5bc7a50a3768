#!/usr/bin/env python3
"""NR PDSCH/PUSCH bit-level receive chain after equalisation (BASELINE config 3 flavour: BG1, Z = 384, 256-QAM):
  soft demodulation (int8) + sign change + descrambling (pdsch_nr.c:456-470)  ->  LDPC rate de-matching  ->  layered min-sum LDPC decoding with CRC early stop (sch_nr.c:606-619)
entirely on the device, and the transmit side (LDPC encoder + rate matching) that produced the test signal.
Prints one JSON line.  Single GPU; equalised symbols resident in HBM."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tbs", type=int, default=1024); ap.add_argument("--cbs-per-tb", type=int, default=8)
    ap.add_argument("--snr", type=float, default=24.0); ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=4)
    a = ap.parse_args()
    import torch
    import srslte_amd as S, oracle_api as O
    from srslte_amd import capi
    lib = S.lib()
    dev = torch.device("cuda", 0)
    capi.check(lib.srsran_hip_set_device(0), "set_device")
    bg, Z, mod, Qm, F = 0, 384, 4, 8, 0
    N, K = 66 * Z, 22 * Z
    E = 12672
    ncb_tb = a.cbs_per_tb
    n_cb = a.tbs * ncb_tb
    nsym_tb = ncb_tb * E // Qm
    st = torch.cuda.current_stream().cuda_stream
    h = C.c_void_p()
    capi.check(lib.srsran_hip_nr_sch_create(C.byref(h)), "nr_sch_create")
    # ---- transmit side on the device: a pool of transport blocks, encoded and rate matched (timed for the record)
    pool_tb = 8
    rng = np.random.default_rng(5)
    msgs = rng.integers(0, 2, (pool_tb * ncb_tb, K)).astype(np.uint8)
    poly, order = 0x1800063, 24  # CRC24B closes every code block (sch_nr.c:443-447), checked after every iteration (:619)
    for m in msgs:
        cs = O.orc().orc_crc_bits(poly, order, O.P(m), K - order)
        m[K - order:] = [(cs >> (order - 1 - j)) & 1 for j in range(order)]
    d_msg = torch.from_numpy(msgs).to(dev)
    d_cw = torch.zeros((pool_tb * ncb_tb, N), dtype=torch.uint8, device=dev)
    d_tx = torch.zeros((pool_tb * ncb_tb, E), dtype=torch.uint8, device=dev)
    encj = (capi.HipLdpcCb * (pool_tb * ncb_tb))(*[capi.HipLdpcCb(i * K, i * N, N) for i in range(pool_tb * ncb_tb)])
    txj = (capi.HipLdpcCb * (pool_tb * ncb_tb))(*[capi.HipLdpcCb(i * N, i * E, E) for i in range(pool_tb * ncb_tb)])
    capi.check(lib.srsran_hip_ldpc_encode_batch(h, d_msg.data_ptr(), d_cw.data_ptr(), encj, pool_tb * ncb_tb, bg, Z, st), "encode")
    capi.check(lib.srsran_hip_ldpc_rm_tx_batch(h, d_cw.data_ptr(), d_tx.data_ptr(), txj, pool_tb * ncb_tb, bg, Z, 0, mod, N, st), "rm_tx")
    torch.cuda.synchronize()
    tx = d_tx.cpu().numpy().reshape(pool_tb, ncb_tb * E)
    par_tx = np.array_equal(tx[0][:E], O.ldpc_rm_tx(O.ldpc_encode_rm(bg, Z, msgs[0], N), E, bg, Z, 0, mod, N))
    seeds = [int(rng.integers(0, 1 << 31)) for _ in range(pool_tb)]
    sigma = 10 ** (-a.snr / 20) / np.sqrt(2)
    syms = []
    for t in range(pool_tb):
        x = O.modulate(tx[t] ^ O.sequence_bits(seeds[t], ncb_tb * E), mod)
        syms.append((x + sigma * (rng.standard_normal(nsym_tb) + 1j * rng.standard_normal(nsym_tb))).astype(np.complex64))
    sym_pool = torch.from_numpy(np.stack(syms).view(np.float32)).to(dev)
    d_sym = sym_pool.repeat((a.tbs + pool_tb - 1) // pool_tb, 1)[:a.tbs].contiguous()
    d_llr = torch.zeros((a.tbs, ncb_tb * E), dtype=torch.int8, device=dev)
    d_soft = torch.zeros((n_cb, N), dtype=torch.int8, device=dev)
    d_out = torch.zeros((n_cb, K), dtype=torch.uint8, device=dev)
    d_nit = torch.zeros(n_cb, dtype=torch.int32, device=dev)
    dj = (capi.HipDemodJob * a.tbs)(*[capi.HipDemodJob(mod, nsym_tb, i * nsym_tb, i * ncb_tb * E, seeds[i % pool_tb], 3) for i in range(a.tbs)])
    rxj = (capi.HipLdpcCb * n_cb)(*[capi.HipLdpcCb(i * E, i * N, E) for i in range(n_cb)])
    new_flags = np.ones(n_cb, np.uint8)
    hd = C.c_void_p()
    capi.check(lib.srsran_hip_demod_create(C.byref(hd)), "demod_create")
    dec = S.LdpcBatch(bg, Z, 0.8, a.iters, n_cb)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    parts = []

    def step():
        # first transmission: the de-matcher gets the new-data flag of every block instead of a cleared soft buffer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        capi.check(lib.srsran_hip_demod_run(hd, d_sym.data_ptr(), d_llr.data_ptr(), capi.LLR_BYTE, dj, a.tbs, st), "demod")
        ev[1].record()
        capi.check(lib.srsran_hip_ldpc_rm_rx_batch_new(h, capi.LLR_BYTE, d_llr.data_ptr(), d_soft.data_ptr(), rxj, new_flags.ctypes.data, n_cb, F, bg, Z, 0, mod, N, st), "rm_rx")
        ev[2].record()
        capi.check(lib.srsran_hip_ldpc_batch_run_crc(dec._h, d_soft.data_ptr(), N, d_out.data_ptr(), K, n_cb, min(E, N), poly, order,
                                                     d_nit.data_ptr(), st), "ldpc_run_crc")
        ev[3].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        parts.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
        return dt

    for _ in range(a.warmup):
        step()
    parts.clear()
    dt = sum(step() for _ in range(a.steps)) / a.steps
    pm = np.mean(np.array(parts), axis=0)
    got = d_out[:pool_tb * ncb_tb].cpu().numpy()
    ok = int((got == msgs).all(axis=1).sum())
    nit = d_nit.cpu().numpy()
    # CPU: oracle restatement of the same chain on a few code blocks of the first transport block (single thread)
    t1 = time.perf_counter()
    llr0 = O.sequence_apply((-O.demod_soft(mod, syms[0], "b").astype(np.int32)).astype(np.int8), seeds[0])  # pdsch_nr.c:456-470
    par = True
    for i in range(min(a.cpu_sample, ncb_tb)):
        soft, n_llr = O.ldpc_rm_rx(llr0[i * E:(i + 1) * E], np.zeros(N, np.int8), F, bg, Z, 0, mod, N)
        out, rets = O.ldpc_decode(bg, Z, soft[None, :], 0.8, a.iters, n_llr, crc=(poly, order))
        par = par and rets[0] == nit[i] and (rets[0] == 0 or np.array_equal(out[0], got[i]))
    tc = time.perf_counter() - t1
    rm_bytes = n_cb * (E + 2 * min(N, max(E, 20 * Z)))
    out = {"metric": "code blocks received, Mbit/s of information bits (256-QAM symbols -> int8 LLRs -> descrambling -> LDPC rate de-matching -> LDPC BG1 Z=384 decoding)",
           "value": n_cb * K / dt / 1e6, "unit": "Mbit/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3,
           "config": {"workload": "%d transport blocks x %d code blocks (BG1, Z=384, E=%d, rate %.2f, 256-QAM), Es/N0 %.1f dB, max %d iterations with CRC24B early stop"
                                  % (a.tbs, ncb_tb, E, K / E, a.snr, a.iters)},
           "code_blocks_recovered_of_pool": [ok, pool_tb * ncb_tb], "crc_ok": int((nit > 0).sum()),
           "avg_iterations": float(nit[nit > 0].mean()) if (nit > 0).any() else 0.0,
           "demod_descramble_ms": float(pm[0]), "rate_dematch_ms": float(pm[1]), "ldpc_decode_ms": float(pm[2]),
           "rate_dematch_GBps": rm_bytes / pm[1] / 1e6,
           "cpu_baseline": {"value": min(a.cpu_sample, ncb_tb) * K / tc / 1e6, "unit": "Mbit/s", "cores": 1, "kind": "port",
                            "sample": "%d code blocks, oracle restatement of demodulate + descramble + rm_rx + decode_c (scalar C)" % min(a.cpu_sample, ncb_tb)},
           "parity_vs_oracle": ("identical iteration counts and decoded bits" if par else "MISMATCH") + ("; device encoder + rate matcher equal the oracle's" if par_tx else "; TX MISMATCH")}
    print(json.dumps(out))


main()
