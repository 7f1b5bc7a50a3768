"""The measured alternatives kept in the tree behind development knobs (DESIGN.md par. 3.2 and 3.4: the turbo decoder's launch
shapes, the three PSS correlation kernels) must stay CORRECT -- a variants table whose rows compute different things says nothing.
They are NOT in the product library: srslte_amd/build.py --variants compiles the three kernel files that hold them with
SRSRAN_HIP_WITH_VARIANTS into tools/probe/lib/libsrsran_phy_hip_variants.so.  The tests below run in a CHILD interpreter that loads that
library (SRSRAN_HIP_LIB); in the parent only the wrapper runs, which also checks that the product library does not carry the kernels.
The knobs' environment variables are read once; srsran_hip_dev_knob overrides them at run time, so one process switches between them."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_api as O
import test_gpu_sync as TS
import test_gpu_turbo as TT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS_LIB = os.path.join(ROOT, "tools", "probe", "lib", "libsrsran_phy_hip_variants.so")
CHILD = os.environ.get("SRSRAN_VARIANTS_CHILD") == "1"
in_child = pytest.mark.skipif(not CHILD, reason="runs in the child interpreter that loads the variants library (test_variants_in_a_child_process)")


@pytest.mark.skipif(CHILD, reason="the wrapper runs in the parent")
def test_variants_in_a_child_process(hiplib):
    assert os.path.exists(VARIANTS_LIB), "tools/probe/lib/libsrsran_phy_hip_variants.so not built (python -m srslte_amd.build --variants)"
    import srslte_amd.capi as capi

    syms = subprocess.run(["nm", "-C", "--defined-only", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    for k in ("tdec_win_kernel_waves1", "tdec_win_kernel_persistent", "pss_pair_kernel", "pss_block_kernel"):
        assert k not in syms, "%s is in the product library" % k
    env = dict(os.environ, SRSRAN_HIP_LIB=VARIANTS_LIB, SRSRAN_VARIANTS_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]
    assert "9 passed" in r.stdout, r.stdout[-800:]  # 6 PSS + 3 turbo cases


@pytest.fixture
def knob(hiplib):
    touched = set()

    def set_(name, value):
        touched.add(name)
        assert hiplib.srsran_hip_dev_knob(name.encode(), value.encode()) == 0

    yield set_
    for name in touched:
        assert hiplib.srsran_hip_dev_knob(name.encode(), None) == 0


@in_child
@pytest.mark.parametrize("variant", ["wave", "pair", "block"])
@pytest.mark.parametrize("N,frame", [(128, 9600), (2048, 61440)])
def test_pss_kernels_agree_with_the_oracle(hiplib, knob, variant, N, frame):
    import srslte_amd as S

    knob("SRSRAN_HIP_PSS_VARIANT", variant)
    prb = 6 if N == 128 else 100
    rng = np.random.default_rng(7 * N)
    cells_id = [3, 151, 302, 500]
    delays = [0] + [int(rng.integers(0, frame - 15 * N)) for _ in cells_id[1:]]
    caps = np.stack([TS._capture(c, prb, N, frame, d, 0.05, rng, sf5=(i % 2 == 1)) for i, (c, d) in enumerate(zip(cells_id, delays))])
    h, got = TS._run_batch(S, caps, frame, N, 1)
    for i, (cid, d) in enumerate(zip(cells_id, delays)):
        for n2 in range(3):
            g = got[i * 3 + n2]
            pk, pv, psr = O.pss_find(caps[i], N, n2)[:3]
            assert g.peak_pos == pk, (variant, cid, n2, g.peak_pos, pk)
            assert abs(g.peak_value - pv) <= 1e-4 * pv and abs(g.psr - psr) <= 1e-3 * psr
        g = got[i * 3 + cid % 3]
        assert g.peak_pos == d + 15 * N // 2 and g.N_id_1 == cid // 3 and g.sf_idx == (5 if i % 2 else 0)
    S.lib().srsran_hip_cellsearch_free(h)


@in_child
@pytest.mark.parametrize("variant", ["product", "waves1", "persistent"])
def test_turbo_launch_shapes_agree_with_the_oracle(hiplib, knob, variant):
    import srslte_amd as S
    from srslte_amd import capi

    knob("SRSRAN_HIP_TDEC_VARIANT", variant)
    # 40 blocks = 5 units of 8: the persistent grid's counter hands out more than one unit
    TT._check(S, 6144, capi.TDEC_AUTO, O.ORC_TDEC_AUTO, 40, 0.0, [1, 4], seed=5)
