"""Test-only host harness (see backend_host.cpp).  Never imported by sanm_amd."""
from __future__ import annotations

import ctypes

from sanm_amd.api import Api

from .build import build

_API = None


def get_hostsim_api() -> Api:
    global _API
    if _API is None:
        lib = ctypes.CDLL(build())
        _API = Api(lib).init(0)
    return _API


_API_NATIVE = None


def get_hostsim_native_api() -> Api:
    """the harness built for speed (-O3 -march=native, contraction allowed): bench.py's cpu_baseline leg only --
    never a parity reference (tests/hostsim/build.py).  Built on first use ON THE BOX IT RUNS ON (-march=native);
    falls back to the parity build where no compiler is available."""
    global _API_NATIVE
    if _API_NATIVE is None:
        try:
            path = build(native=True)
        except Exception:  # noqa: BLE001 -- no g++ on the box: the strict build travels with the repository
            path = build()
        _API_NATIVE = Api(ctypes.CDLL(path)).init(0)
    return _API_NATIVE
