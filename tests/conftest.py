import os
import sys

import pytest

try:  # torch must load its bundled HIP runtime before libsanm_hip.so loads the system one
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _get_api(kind):
    if kind == "hip":
        import sanm_amd
        return sanm_amd.get_api()
    from tests.hostsim import get_hostsim_api
    return get_hostsim_api()


# Every parity test runs twice: on the real HIP product library (-m gpu) and on
# the test-only host harness (same C++ host code + same per-tet bodies, CPU
# loops; see tests/hostsim/backend_host.cpp), which is what the GPU-less
# authoring container can execute.
@pytest.fixture(scope="session", params=["hostsim", pytest.param("hip", marks=pytest.mark.gpu)])
def api(request):
    return _get_api(request.param)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
