"""Build oracle/_ref/libref_poly_<variant>.so from the REFERENCE's own sources.

ORACLE -- test infrastructure only (see oracle/__init__.py).

The reference as a whole is unbuildable in this image (Eigen, MKL headers, Catch2 and nlohmann-json are absent:
SURVEY.md section 8c), but its scalar polynomial translation unit is self-contained:

    /root/reference/libsanm/unary_polynomial.cpp   (eval, solve_eqn, roots = ACM algorithm 30, minimize ...)
    /root/reference/libsanm/utils.cpp              (sanm_assert, ssprintf)
    /root/reference/third_party/BRENT/brent.cpp    (vendored Brent zero / glomin)

They are compiled from where they lie (never copied into this repository) together with oracle/ref_wrap.cpp, our
extern "C" wrapper.  `unary_polynomial::eval_tensor` references two TensorND operators that live in an
Eigen-dependent file; nothing the wrapper exports calls it, and `-ffunction-sections -Wl,--gc-sections` with hidden
visibility drops it at link time, so the library has no unresolved symbol and no stand-in is written for anything.

Variants (the root finder is sensitive to floating-point contraction on the ill-scaled Pade denominators):
  O2      g++ -O2                                  strict IEEE double evaluation of the source, no FMA contraction
                                                   -- what the fixtures and the restatements pin
  native  g++ -O2 -g -DNDEBUG -march=native        the reference's own flags (CMakeLists.txt:5-12, RelWithDebInfo);
                                                   machine dependent: gcc contracts a*b+c into FMA where the CPU has it

Outputs: oracle/_ref/ (git-ignored and, since round 4, gpurun-ignored: nothing on the GPU box loads it).  /root/reference does not exist on the GPU box; nothing at
run time needs these libraries there (tests that use them skip when they are absent).

Usage: python -m oracle.build_ref [--force]
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SANM_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
WRAP = os.path.join(HERE, "ref_wrap.cpp")
REF_SOURCES = ["libsanm/unary_polynomial.cpp", "libsanm/utils.cpp", "third_party/BRENT/brent.cpp"]
VARIANTS = {
    "O2": ["-O2"],
    "native": ["-O2", "-g", "-DNDEBUG", "-march=native"],
}
COMMON = ["-std=c++20", "-fPIC", "-shared", "-fvisibility=hidden", "-ffunction-sections", "-Wl,--gc-sections"]


def lib_path(variant: str = "O2") -> str:
    return os.path.join(OUT_DIR, f"libref_poly_{variant}.so")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "libsanm"))


def build(force: bool = False) -> list[str]:
    """Compile every variant; returns the paths built (empty when the reference tree is absent)."""
    if not available():
        return []
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = [os.path.join(REF, s) for s in REF_SOURCES]
    outs = []
    for variant, flags in VARIANTS.items():
        out = lib_path(variant)
        outs.append(out)
        if not force and os.path.exists(out) and all(os.path.getmtime(s) <= os.path.getmtime(out) for s in srcs + [WRAP]):
            continue
        cmd = ["g++"] + COMMON + flags + ["-I", REF, "-I", os.path.join(REF, "third_party/BRENT"), WRAP] + srcs + ["-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference build failed: %s\n%s" % (" ".join(cmd), r.stderr))
    return outs


class RefPoly:
    """ctypes view of one variant of the reference's unary_polynomial functions."""

    def __init__(self, variant: str = "O2"):
        path = lib_path(variant)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.ref_poly_roots.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, dp, ctypes.c_int]
        lib.ref_poly_roots.restype = ctypes.c_int
        lib.ref_poly_eval.argtypes = [dp, ctypes.c_int, ctypes.c_double]
        lib.ref_poly_eval.restype = ctypes.c_double
        lib.ref_poly_solve_eqn.argtypes = [dp, ctypes.c_int] + [ctypes.c_double] * 4 + [ctypes.POINTER(ctypes.c_int)]
        lib.ref_poly_solve_eqn.restype = ctypes.c_double
        lib.ref_poly_stable_x_range.argtypes = [ctypes.c_int]
        lib.ref_poly_stable_x_range.restype = ctypes.c_double
        self.lib = lib
        self.variant = variant

    @staticmethod
    def _arr(f):
        f = [float(v) for v in f]
        return (ctypes.c_double * len(f))(*f), len(f)

    def roots(self, f, only_real=True, max_iter=300, tol=1e-8):
        """unary_polynomial::roots; list of complex, or None where the reference returns None."""
        a, n = self._arr(f)
        re = (ctypes.c_double * (n + 2))()
        im = (ctypes.c_double * (n + 2))()
        k = self.lib.ref_poly_roots(a, n, int(only_real), max_iter, tol, re, im, n + 2)
        if k == -1:
            return None
        if k < 0:
            raise AssertionError("reference assertion in roots()")
        return [complex(re[i], im[i]) for i in range(k)]

    def eval(self, f, x):
        a, n = self._arr(f)
        return self.lib.ref_poly_eval(a, n, float(x))

    def solve_eqn(self, f, xmin, xmax, b=0.0, eps=1e-6):
        a, n = self._arr(f)
        ok = ctypes.c_int(0)
        r = self.lib.ref_poly_solve_eqn(a, n, xmin, xmax, b, eps, ctypes.byref(ok))
        if not ok.value:
            raise AssertionError("reference assertion in solve_eqn()")
        return r

    def stable_x_range(self, order):
        return self.lib.ref_poly_stable_x_range(int(order))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
