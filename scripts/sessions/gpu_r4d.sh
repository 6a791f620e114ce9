#!/bin/bash
# round 4, session d: whole GPU suite on the new code, finer leaf sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4d
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
LEAVES="12 16 20 24 28 32" MERGES="1 1,3" bash scripts/sweep_leaf_merge.sh r4d_sweep > /dev/null 2>&1
cat $ROOT/gpurun_out/r4d_sweep/sweep.md
LEAVES="16 32" MERGES="none" WORKLOADS="block:32" bash scripts/sweep_leaf_merge.sh r4d_sweep_block32 > /dev/null 2>&1
cat $ROOT/gpurun_out/r4d_sweep_block32/sweep.md
