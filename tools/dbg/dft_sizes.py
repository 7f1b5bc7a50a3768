"""DFT sizes the sidelink / NB-IoT objects of the reference plan (2 x 15 x symbol size), all option sets, against numpy float64"""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle_api as O
import srslte_amd as S
from srslte_amd import capi
lib = S.lib()
lib.srsran_hip_set_device(0)
for n in [int(a) for a in sys.argv[1:]] or [3840, 7680, 11520, 15360, 23040, 30720, 46080, 61440]:
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    for backward in (0, 1):
        plan = capi.DftPlan()
        rc = lib.srsran_dft_plan(C.byref(plan), n, backward, 0)
        if rc:
            print(n, backward, "plan failed", S.capi.last_error()); continue
        for mirror, dc, norm in ((0, 0, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)):
            lib.srsran_dft_plan_set_mirror(C.byref(plan), bool(mirror)); lib.srsran_dft_plan_set_dc(C.byref(plan), bool(dc)); lib.srsran_dft_plan_set_norm(C.byref(plan), bool(norm))
            y = np.zeros(n, np.complex64)
            lib.srsran_dft_run_c(C.byref(plan), O.P(x), O.P(y))
            ref = np.zeros(n, np.complex64)
            O.orc().orc_dft_c(O.P(x), O.P(ref), n, backward, mirror, dc, norm)
            e = float(np.abs(y - ref).max()) / max(1.0, float(np.sqrt(np.mean(np.abs(ref) ** 2))))
            print(n, "bwd" if backward else "fwd", (mirror, dc, norm), "err %.2e" % e, "" if e < 1e-4 else "  <<<<<<")
        lib.srsran_dft_plan_free(C.byref(plan))
