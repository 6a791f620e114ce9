"""Run-time specialisation of the Taylor pass kernels: the generated source must compile for gfx950 (no GPU
needed: hiprtc is only a compiler here); that the compiled kernels compute the same as the interpreter kernels
is a GPU test (test_device_anm.py runs every model both ways)."""
import ctypes as C
import os

import numpy as np
import pytest

import sanm_amd
from sanm_amd import fea

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _product_lib():
    path = os.path.join(ROOT, "sanm_amd", "libsanm_hip.so")
    if not os.path.exists(path):
        pytest.skip("libsanm_hip.so not built")
    lib = C.CDLL(path)
    lib.sanm_rtc_compile_check.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    return lib


def _check(lib, src):
    log = C.create_string_buffer(1 << 16)
    size = C.c_size_t()
    rc = lib.sanm_rtc_compile_check(src.encode(), log, len(log), C.byref(size))
    assert rc == 0, log.value.decode()[-4000:]
    return size.value


def test_headers_compile_at_run_time():
    lib = _product_lib()
    src = '#include "tet_ops.h"\nusing namespace sanm_hip;\n' \
          'extern "C" __global__ void k(ProgramDev P, int order, const double* x) {\n' \
          '  extern __shared__ double cur[];\n' \
          '  exec_program_tet(P, PASS_COEFF, order, blockIdx.x * 64 + (threadIdx.x & 63), x, cur + (threadIdx.x & 63), 64);\n}\n'
    assert _check(lib, src) > 1000


@pytest.mark.parametrize("energy", ["neohookean_c", "neohookean_i", "arap"])
def test_generated_source_compiles(energy):
    """the source generated for a compiled graph (taken from the host harness, which shares graph.cpp)"""
    from tests.hostsim import get_hostsim_api
    api = get_hostsim_api()
    lib = _product_lib()
    cfg = {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": energy, "order": 6}
    run = fea.GravityRun(api, fea.make_cuboid(3, 3, 4, 0.025), cfg).construct()
    src = run.solver.spec_source()
    assert "spec_pass3" in src and "SPEC_OPS" in src
    assert _check(lib, src) > 10000


def test_code_object_cache(tmp_path, monkeypatch):
    """rtc.cpp: a code object is compiled once; the second request of the same source is served from the process
    cache and, in a new process, from the on-disk cache ($SANM_JIT_CACHE_DIR).  No GPU needed (hiprtc only compiles);
    run in subprocesses because the process cache is what is being observed."""
    import subprocess
    import sys
    prog = r'''
import ctypes as C, os, sys
lib = C.CDLL(os.path.join(%r, "sanm_amd", "libsanm_hip.so"))
lib.sanm_rtc_cache_probe.argtypes = [C.c_char_p]
src = b'#include "program.h"\nextern "C" __global__ void k(double* x) { x[threadIdx.x] = %s; }\n'
n = int(sys.argv[1])
for _ in range(n):
    assert lib.sanm_rtc_cache_probe(src) == 0
c, m, d = C.c_int64(), C.c_int64(), C.c_int64()
lib.sanm_rtc_cache_stats(C.byref(c), C.byref(m), C.byref(d))
print(c.value, m.value, d.value)
''' % (ROOT, "1.5")
    env = dict(os.environ, SANM_JIT_CACHE_DIR=str(tmp_path / "cache"))
    run = lambda n, e=env: subprocess.run([sys.executable, "-c", prog, str(n)], env=e, capture_output=True, text=True,
                                          check=True).stdout.split()
    assert run(3) == ["1", "2", "0"]          # compiled once, twice from memory
    files = os.listdir(tmp_path / "cache")
    assert len(files) == 1 and files[0].endswith(".hsaco")
    assert run(2) == ["0", "1", "1"]          # a new process: from disk, then from memory
    env2 = dict(env, SANM_NO_JIT_CACHE="1")
    assert run(1, env2) == ["1", "0", "0"]    # disk cache off: compiled again
