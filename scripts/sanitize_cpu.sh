#!/bin/bash
# The host harness (the same C++ host code and operator bodies as the product, tests/hostsim) built with
# AddressSanitizer + UndefinedBehaviorSanitizer and the CPU test suite run against it.  (GPU sanitizers are not
# available on the pool; the shared bodies -- tet_ops.h, vecprog.h, row_ops.h -- and all host logic are covered here.)
# usage: bash scripts/sanitize_cpu.sh [pytest args]
set -eu
cd "$(dirname "$0")/.."
python - <<'PY'
import importlib.util, os, subprocess
spec = importlib.util.spec_from_file_location("hb", "tests/hostsim/build.py")
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES] + [os.path.join(B.HERE, "backend_host.cpp"), os.path.join(B.HERE, "pardiso_solver.cpp")]
cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-mfma", "-std=c++20",
       "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I", B.CSRC, "-o", B.OUT] + srcs + ["-ldl"]
subprocess.run(cmd, check=True)
PY
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests -q -m "not gpu" --deselect tests/test_bench_multiproc.py --deselect tests/test_sharded.py "$@" || rc=$?
python tests/hostsim/build.py --force > /dev/null   # back to the normal build
exit ${rc:-0}
