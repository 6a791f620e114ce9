"""adapter/anm_hip.h LINKED AND RUN against the reference's own graph code, in the authoring
container only (it needs /root/reference; skipped elsewhere).

The reference's graph layer -- libsanm/{utils,symbolic,oprs,analytic_unary}.cpp and oprs/{elem_arith,linalg,misc,
reduce}.cpp -- is Eigen-free and is compiled from where it lies; its numerical layer (tensor*.cpp, oprs/
analytic_unary.cpp: Eigen) is not.  Every symbol the link then misses gets an ABORTING stub, generated from the
linker's own list of unresolved mangled names: no reference code is re-implemented and none of its arithmetic runs
(the one exception to "abort": an unresolved `...::instance()` returns NULL, so that OperatorNode::isinstance<> of a
meta class that is not linked is simply false).  tests/adapter_link/driver.cpp builds graphs with the reference's
SymbolVar API, sanm::hip::export_graph walks the real OperatorNodes / metas / private Param records, and the exported
sanm_graph is evaluated through the C ABI of the host harness (same capi.cpp as the product).  The same graphs built
through the Python binding must give the same numbers: the adapter's export is then the graph the reference holds.
Nothing of the reference is copied; the objects live in a temporary directory."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference"
REF_TUS = ["libsanm/utils.cpp", "libsanm/symbolic.cpp", "libsanm/oprs.cpp", "libsanm/analytic_unary.cpp",
           "libsanm/oprs/elem_arith.cpp", "libsanm/oprs/linalg.cpp", "libsanm/oprs/misc.cpp", "libsanm/oprs/reduce.cpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "libsanm")), reason="the reference tree is not on this box")


def _sh(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    assert r.returncode == 0, (" ".join(cmd), r.stderr[-3000:])
    return r.stdout


def _build(tmp):
    from tests.hostsim import get_hostsim_api
    get_hostsim_api()
    libdir = os.path.join(ROOT, "tests", "hostsim")
    objs = []
    for tu in REF_TUS:
        o = os.path.join(tmp, tu.replace("/", "_") + ".o")
        _sh(["g++", "-std=c++20", "-O1", "-fPIC", "-I", REF, "-c", os.path.join(REF, tu), "-o", o])
        objs.append(o)
    drv = os.path.join(tmp, "driver.o")
    _sh(["g++", "-std=c++20", "-O1", "-fPIC", "-Wall", "-I", REF, "-I", os.path.join(ROOT, "include"), "-I",
         os.path.join(ROOT, "adapter"), "-c", os.path.join(ROOT, "tests", "adapter_link", "driver.cpp"), "-o", drv])
    objs.append(drv)
    # what the objects need and do not define themselves, by mangled name; keep the reference's own (namespace sanm)
    und = set(_sh(["nm", "--undefined-only", "-P"] + objs).split()[0::2] if False else
              [ln.split()[0] for ln in _sh(["nm", "--undefined-only", "-P"] + objs).splitlines() if ln and not ln.endswith(":")])
    dfn = set(ln.split()[0] for ln in _sh(["nm", "--defined-only", "-P"] + objs).splitlines() if ln and not ln.endswith(":"))
    missing = sorted(s for s in und - dfn if "4sanm" in s and not s.startswith("sanm_"))
    funcs = [s for s in missing if not s.startswith(("_ZTV", "_ZTI", "_ZTS"))]
    assert len(funcs) == len(missing), [s for s in missing if s not in funcs]  # (no data symbol may be missing)
    stub = os.path.join(tmp, "stubs.c")
    with open(stub, "w") as f:
        f.write("#include <stdio.h>\n#include <stdlib.h>\n")
        for s in funcs:
            if s.endswith("8instanceEv"):
                f.write(f"void* {s}(void) {{ return 0; }}\n")
            else:
                f.write(f'void* {s}(void) {{ fprintf(stderr, "aborting stub called: {s}\\n"); abort(); }}\n')
    so = os.path.join(tmp, "stubs.o")
    _sh(["gcc", "-O0", "-fPIC", "-c", stub, "-o", so])
    exe = os.path.join(tmp, "adapter_link")
    _sh(["g++", "-o", exe] + objs + [so, "-L", libdir, "-l:libsanm_hostsim.so", f"-Wl,-rpath,{libdir}", "-pthread"])
    return exe, len(funcs)


def _ours(api, name, T, order, xs):
    """the same graph through the Python binding, evaluated like the driver does"""
    from sanm_amd import api as A
    import scipy.sparse as sp
    g = A.ComputingGraph(api)
    F = g.placeholder()
    mu, lam = 1.25, 0.75
    if name == "arap":
        P = (F - F.batched_svd_w(True)[2]) * mu
    elif name == "stvk":
        P = A.linear_combine([(mu, F.batched_matmul(F.batched_transpose()).batched_matmul(F)), (-mu, F)])
    else:
        FTinv = A.batched_mat_inv_mul(F, None, True).batched_transpose()
        J = F.batched_det()
        Ic = (F * F).reduce_sum(-1)
        t2 = A.linear_combine([(mu / -3.0, J * Ic), (lam, J * J), (-lam, J)], 0.5) * FTinv
        P = A.linear_combine([(mu, J * F), (1.0, t2), (0.25, Ic.batched_mul_eye(3))])
    prop = A.TaylorCoeffProp(api, P, A.SparseLinearDesc(api, sp.identity(T * 9, format="csr")), order, T)
    out = {("y", 0): prop.push_xi(xs[0]).ravel()}
    out[("jac", 0)] = prop.get_jacobian().ravel()
    for k in range(1, order + 1):
        out[("bias", k)] = prop.compute_next_order_bias().ravel()
        out[("y", k)] = prop.push_xi(xs[k]).ravel()
    return out


def test_export_graph_runs_on_the_references_operator_nodes(tmp_path):
    from tests.hostsim import get_hostsim_api
    api = get_hostsim_api()
    exe, nstub = _build(str(tmp_path))
    T, order = 5, 4
    rng = np.random.default_rng(11)
    xs = [np.tile(np.eye(3).ravel(), T) + 0.1 * rng.standard_normal(T * 9)] + [0.3 * rng.standard_normal(T * 9) for _ in range(order)]
    inp = tmp_path / "x.txt"
    inp.write_text("\n".join(" ".join(repr(float(v)) for v in x) for x in xs))
    r = subprocess.run([exe, str(T), str(order), str(inp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    got, name = {}, None
    for ln in r.stdout.splitlines():
        tok = ln.split()
        if tok[0] == "graph":
            name = tok[1]
            got[name] = {}
        else:
            got[name][(tok[0], int(tok[1]))] = np.array([float(v) for v in tok[2:]])
    assert set(got) == {"arap", "stvk", "nh_parts"}
    for name, vals in got.items():
        ref = _ours(api, name, T, order, xs)
        assert set(vals) == set(ref)
        for key in ref:
            a, b = vals[key], ref[key]
            assert a.shape == b.shape and np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max()), (name, key)
    print(f"adapter linked against {len(REF_TUS)} reference translation units ({nstub} aborting stubs for the Eigen "
          f"layer); 3 graphs exported and evaluated: {sorted(got)}")
