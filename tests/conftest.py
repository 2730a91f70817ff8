import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def pytest_configure(config):
    """LOCATOR_HIP_LIB=<path>: run the suite against another build of the library (tests only; the library itself reads
    no environment): e.g. `make -C locator_amd/csrc debug_drain` -> locator_amd/liblocator_hip_drain.so, whose GEMM kernels
    drain every load where the product build counts them by hand."""
    import os
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    alt = os.environ.get("LOCATOR_HIP_LIB")
    if alt:
        from locator_amd import _lib
        _lib.use_library(os.path.abspath(alt))
