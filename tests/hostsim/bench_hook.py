"""Test-only hook for bench.py (SANM_BENCH_TEST_HOOK=tests.hostsim.bench_hook): the host harness in place of the HIP
library, so that the N > 1 launcher of `python bench.py --gpus N` can be exercised in the GPU-less container.  The
bench line carries "backend": "hostsim" when this ran -- it is never a measurement."""
from tests.hostsim import get_hostsim_api


def install(bench):
    bench.make_api = lambda local_rank: get_hostsim_api()
    bench.device_sync = lambda: None
