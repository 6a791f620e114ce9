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
