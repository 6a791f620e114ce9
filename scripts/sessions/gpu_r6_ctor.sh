#!/bin/bash
# after constructor changes: the solver / driver / distribution GPU tests, then the default bench line      usage: gpu_r6_ctor.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests/test_direct_solver.py tests/test_device_anm.py tests/test_gpu_dist.py tests/test_tikhonov.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $OUT/pytest.log | tail -2
bash scripts/sessions/gpu_r6_final.sh $TAG bench
