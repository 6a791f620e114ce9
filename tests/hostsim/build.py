"""Build the TEST-ONLY host harness tests/hostsim/libsanm_hostsim.so.

Same host sources as the product (graph compilation, assembly pattern, ANM
driver, Pade, fea, C ABI) but linked with tests/hostsim/backend_host.cpp
instead of the HIP backend, compiled by g++ (no HIP at all).  Never loaded by
the sanm_amd package.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.abspath(os.path.join(HERE, "..", "..", "sanm_amd", "csrc"))
OUT = os.path.join(HERE, "libsanm_hostsim.so")
SOURCES = ["graph.cpp", "vecprog.cpp", "sparse.cpp", "backend_common.cpp", "poly.cpp", "anm.cpp", "multifrontal.cpp", "fea.cpp", "capi.cpp"]


def build(force=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "backend_host.cpp"),
                                                         os.path.join(HERE, "pardiso_solver.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    cmd = ["g++", "-O2", "-ffp-contract=off", "-mfma", "-std=c++20", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-Wall",
           "-Wno-unused-function", "-I", CSRC, "-o", OUT] + srcs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hostsim build failed:\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
