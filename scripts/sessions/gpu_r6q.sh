#!/bin/bash
# reproduce the abort of the full-size suite     usage: gpu_r6q.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
echo "== fullsize alone"; timeout 1200 python -X faulthandler -m pytest tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/a.log 2>&1; echo "rc=$?"; tail -3 $OUT/a.log | cut -c1-200
echo "== fullsize alone, box grids"; SANM_MF_SOLVE_LISTS=0 timeout 1200 python -X faulthandler -m pytest tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/b.log 2>&1; echo "rc=$?"; tail -3 $OUT/b.log | cut -c1-200
echo "== fullsize alone, lists, LS_WIDTH=0"; SANM_MF_LS_WIDTH=0 timeout 1200 python -X faulthandler -m pytest tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/c.log 2>&1; echo "rc=$?"; tail -3 $OUT/c.log | cut -c1-200
grep -l "Aborted\|fault" $OUT/*.log
for f in $OUT/a.log $OUT/b.log $OUT/c.log; do echo "--- $f"; grep -n "Aborted\|fault\|Error\|error" $f | head -5; done
