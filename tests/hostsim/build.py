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
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.abspath(os.path.join(HERE, "..", "..", "sanm_amd", "csrc"))
OUT = os.path.join(HERE, "libsanm_hostsim.so")
# The same harness built for SPEED: bench.py's cpu_baseline leg only.  The reference builds with `-O2 -g -DNDEBUG
# -march=native` (CMakeLists.txt:5-12, RelWithDebInfo); this variant takes -O3 -march=native and lets the compiler
# contract a*b+c again (the parity build above must not: -ffp-contract=off keeps its arithmetic the device's).
# It is made on the box that runs the baseline (-march=native) and does not travel (.gpurunignore).
OUT_NATIVE = os.path.join(HERE, "libsanm_hostsim_native.so")
SOURCES = ["graph.cpp", "vecprog.cpp", "sparse.cpp", "backend_common.cpp", "poly.cpp", "anm.cpp", "multifrontal.cpp", "fea.cpp", "capi.cpp"]


def build(force=False, native=False):
    out = OUT_NATIVE if native else OUT
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "backend_host.cpp"),
                                                         os.path.join(HERE, "pardiso_solver.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    opt = ["-O3", "-march=native"] if native else ["-O2", "-ffp-contract=off", "-mfma"]
    common = opt + ["-std=c++20", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function", "-I", CSRC]
    # one object per source, compiled side by side
    objdir = os.path.join(HERE, "build_native" if native else "build")
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]

    def cc(pair):
        src, obj = pair
        r = subprocess.run(["g++"] + common + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hostsim build failed:\n" + r.stderr)

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    r = subprocess.run(["g++", "-shared", "-pthread", "-Wl,-Bsymbolic", "-o", out] + objs + ["-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hostsim build failed:\n" + r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, native="--native" in sys.argv))
