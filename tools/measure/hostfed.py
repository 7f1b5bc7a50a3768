#!/usr/bin/env python3
"""host-fed turbo decoding over chunk sizes / stream counts (HOSTFED_CHUNK, HOSTFED_STREAMS)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import srslte_amd as S, oracle_api as O, bench_legs as L
from srslte_amd import capi
S.lib().srsran_hip_set_device(0)
dev = torch.device("cuda", 0)
K = 6144
_, pool = O.turbo_llrs(K, 64, 1.0, seed=1)
d_llr = torch.from_numpy(pool).to(dev).repeat(1024, 1)
for chunk, ns in ((16380, 3), (4095, 4), (8190, 4)):
    os.environ["HOSTFED_CHUNK"], os.environ["HOSTFED_STREAMS"] = str(chunk), str(ns)
    for llr8 in (False, True):
        r = L.host_fed_turbo(S, capi, torch, dev, d_llr, 3 * K + 12, K, 8, llr8)
        print(chunk, ns, "int8" if llr8 else "int16", "%.0f Mbit/s, one stream %.0f, h2d %.1f GB/s, kernel alone %.0f, bound %.0f (%s) frac %.2f" % (r["value"], r["one_stream_no_overlap_mbit_per_s"], r["h2d_alone_gb_per_s"], r["kernel_alone_mbit_per_s"], r["bound_mbit_per_s"], r["bound"], r["frac_of_bound"]), flush=True)
