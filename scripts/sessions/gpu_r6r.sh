#!/bin/bash
# is the abort of the full suite reproducible, and does it follow the solve lists?     usage: gpu_r6r.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
echo "== full suite, default"; timeout 1800 python -X faulthandler -m pytest tests -q -m gpu -x > $OUT/a.log 2>&1; echo "rc=$?"; grep -v "^  File\|^Extension" $OUT/a.log | tail -6 | cut -c1-300
echo "== full suite, box grids"; SANM_MF_SOLVE_LISTS=0 timeout 1800 python -X faulthandler -m pytest tests -q -m gpu -x > $OUT/b.log 2>&1; echo "rc=$?"; grep -v "^  File\|^Extension" $OUT/b.log | tail -6 | cut -c1-300
echo "== gpu_dist + fullsize, default, glibc checks"; MALLOC_CHECK_=3 timeout 1800 python -X faulthandler -m pytest tests/test_gpu_dist.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/c.log 2>&1; echo "rc=$?"; grep -v "^  File\|^Extension" $OUT/c.log | tail -6 | cut -c1-300
