"""sanm_amd: MI355X-native ANM hot path of jia-kai/SANM.

The package is a thin ctypes layer over ``libsanm_hip.so`` (hand-written HIP
for gfx950 behind the C ABI in ``include/sanm_hip.h``).  There is no CPU
fallback: loading fails loudly if the library has not been built and
``get_api()`` fails if no HIP device is present.
"""
from __future__ import annotations

import ctypes
import os
import sys

from . import api as _api
from .api import (ANMEqnSolver, ANMImplicitSolver, ANMSolverVecScale, Api, SanmError,  # noqa: F401
                  SanmAssertionError, SanmNumericalError, SanmUnsupportedError, SymbolVar,
                  TaylorCoeffProp, batched_mat_inv_mul, concat, constant, linear_combine, placeholder)

# (SANM_HIP_LIBRARY: another build of the same library, for A/B measurements -- scripts/build_variants.py)
LIB_PATH = os.environ.get("SANM_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsanm_hip.so")
_API = None


def load_library() -> ctypes.CDLL:
    """dlopen the HIP product library (no device needed for this step)."""
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m sanm_amd.build` (needs hipcc). "
            "sanm_amd has no CPU fallback.")
    # A process that uses PyTorch-ROCm as well (bench.py, the distributed tests) must end up with ONE HIP runtime: the
    # wheel bundles its own libamdhip64 / libhsa-runtime64, and if this library pulls in the system's copies first, a later
    # torch.cuda initialisation reports "No HIP GPUs are available" (seen on the MI355X boxes).  With torch loaded first
    # both bind to its copies.  SANM_NO_TORCH_PRELOAD=1: leave the order to the caller.
    if "torch" not in sys.modules and not os.environ.get("SANM_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:  # no torch in this environment: nothing to keep consistent
            pass
    return ctypes.CDLL(LIB_PATH)


def get_api(device: int = 0) -> Api:
    """Load the library and bind HIP device ``device``; raises if there is no GPU."""
    global _API
    if _API is None:
        a = Api(load_library())
        a.init(device)
        a.device = device
        _API = a
    elif getattr(_API, "device", device) != device:
        raise RuntimeError(f"the library is bound to device {_API.device} (one process per GPU); "
                           f"device {device} was requested")
    return _API
