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
