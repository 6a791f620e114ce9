"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Intel MKL PARDISO through ctypes -- the very solver the reference factorises its
Jacobian with (libsanm/sparse_solver.cpp:88-127: `pardisoinit` for mtype 11, zero-based
CSR, phases 12 / 33 / -1).  The image ships MKL 2021.4 in /opt/conda/lib without headers
(SURVEY.md 8c); the prototypes below are MKL's documented C interface

    void pardisoinit(void* pt[64], const MKL_INT* mtype, MKL_INT iparm[64]);
    void pardiso(void* pt[64], const MKL_INT* maxfct, const MKL_INT* mnum, const MKL_INT* mtype,
                 const MKL_INT* phase, const MKL_INT* n, const void* a, const MKL_INT* ia,
                 const MKL_INT* ja, MKL_INT* perm, const MKL_INT* nrhs, MKL_INT* iparm,
                 const MKL_INT* msglvl, void* b, void* x, MKL_INT* error);

with MKL_INT = int32 (the LP64 interface the reference links, libsanm/CMakeLists.txt:41-43).
When the library is absent `available()` is False and oracle.anm falls back to SuperLU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB = None
_TRIED = False


def _load():
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    if os.environ.get("SANM_ORACLE_SOLVER", "").lower() == "superlu":
        return None
    for path in ("/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "libmkl_rt.so"):
        try:
            lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
            lib.pardiso, lib.pardisoinit  # noqa: B018  (symbols must exist)
            # one thread: the oracle must be deterministic (the reference switches to the
            # parallel nested dissection, iparm[1] = 3, only when it runs threaded)
            try:
                lib.MKL_Set_Num_Threads(C.c_int(1))
            except AttributeError:
                pass
            _LIB = lib
            break
        except (OSError, AttributeError):
            continue
    return _LIB


def available() -> bool:
    return _load() is not None


class Pardiso:
    """One real nonsymmetric system (mtype 11): factor once, solve many."""

    def __init__(self, A_csr):
        lib = _load()
        assert lib is not None
        A = A_csr.tocsr()
        A.sort_indices()
        self._lib = lib
        self.n = C.c_int32(A.shape[0])
        self.a = np.ascontiguousarray(A.data, dtype=np.float64)
        self.ia = np.ascontiguousarray(A.indptr, dtype=np.int32)
        self.ja = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.pt = (C.c_void_p * 64)()
        self.iparm = (C.c_int32 * 64)()
        self.mtype = C.c_int32(11)
        lib.pardisoinit(self.pt, C.byref(self.mtype), self.iparm)
        self.iparm[17] = 0  # sparse_solver.cpp:120-122
        self.iparm[18] = 0
        self.iparm[34] = 1  # zero-based indexing
        self._factored = False
        self._call(12, None, None)  # analysis + numerical factorisation (sparse_solver.cpp:336)
        self._factored = True

    def _call(self, phase, b, x):
        one, zero, err = C.c_int32(1), C.c_int32(0), C.c_int32(0)
        self._lib.pardiso(self.pt, C.byref(one), C.byref(one), C.byref(self.mtype), C.byref(C.c_int32(phase)),
                          C.byref(self.n), self.a.ctypes.data_as(C.c_void_p),
                          self.ia.ctypes.data_as(C.POINTER(C.c_int32)),
                          self.ja.ctypes.data_as(C.POINTER(C.c_int32)), None, C.byref(one), self.iparm,
                          C.byref(zero), None if b is None else b.ctypes.data_as(C.c_void_p),
                          None if x is None else x.ctypes.data_as(C.c_void_p), C.byref(err))
        assert err.value == 0, f"pardiso phase={phase} failed: error={err.value}"

    def solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty_like(b)
        self._call(33, b, x)
        return x

    def __del__(self):
        if getattr(self, "_factored", False):
            try:
                self._call(-1, None, None)
            except Exception:  # interpreter shutdown
                pass
