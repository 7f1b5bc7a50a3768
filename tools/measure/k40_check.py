#!/usr/bin/env python3
"""K = 40 blocks at the noise level of turbodecoder_test -e 6.0 (LLR = 100 (+-1 + 0.868 n)), 10 half iterations, one call at a time through the
handle API (queue) and through the batch API: every output against the oracle's scalar decoder"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srslte_amd as S, oracle_api as O
from srslte_amd import capi
lib = S.lib(); lib.srsran_hip_set_device(0)
K, nit, N = int(sys.argv[1]) if len(sys.argv) > 1 else 40, 10, 600
rng = np.random.default_rng(5)
bits = rng.integers(0, 2, (N, K)).astype(np.uint8)
enc = np.stack([O.turbo_encode(b) for b in bits])
llr = np.clip(np.round(100 * ((2.0 * enc - 1) + 0.868 * rng.standard_normal(enc.shape))), -32768, 32767).astype(np.int16)
ref = O.turbo_decode(llr, nit, K)
h = capi.Tdec()
assert lib.srsran_tdec_init(C.byref(h), 6144) == 0
lib.srsran_tdec_force_not_sb(C.byref(h))
bad = []
out = np.zeros(K // 8, np.uint8)
for rep in range(3):
    for i in range(N):
        x = llr[i].copy()
        assert lib.srsran_tdec_run_all(C.byref(h), O.P(x), O.P(out), nit, K) == 0
        if not np.array_equal(out, ref[i]):
            bad.append((rep, i, int(np.unpackbits(out ^ ref[i]).sum())))
print("handle API: %d mismatches of %d" % (len(bad), 3 * N), bad[:10])
got = S.TdecBatch(K, N, capi.TDEC_AUTO).decode(llr, nit)
print("batch API: %d mismatching blocks of %d" % (int((got != ref).any(axis=1).sum()), N))
