#!/bin/bash
# the full GPU suite, twice, on the last tree (the one-off abort of session r6p)     usage: gpu_r6t.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
ulimit -c 0
for i in 1 2; do
  timeout 1800 python -m pytest tests -q -m gpu -x > $OUT/run$i.log 2>&1; echo "run $i rc=$?"; grep -v "^  File\|^Extension" $OUT/run$i.log | tail -2 | cut -c1-200
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
