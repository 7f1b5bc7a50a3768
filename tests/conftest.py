import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hiplib():
    """The product library.  GPU tests FAIL (not skip) when it or the device is missing: no fallback."""
    import srslte_amd as S

    lib = S.lib()
    assert lib.srsran_hip_device_count() > 0, "no HIP device visible: %s" % S.capi.last_error()
    S.capi.check(lib.srsran_hip_set_device(0), "set_device")
    return lib
