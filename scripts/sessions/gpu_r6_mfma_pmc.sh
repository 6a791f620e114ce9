#!/bin/bash
# matrix-core busy cycles of the factorisation's kernels by PMC (one pass; --kernel-trace only)   usage: gpu_r6_mfma_pmc.sh <tag> [workload]
set -u
TAG=$1; WL=${2:-refine:armadillo_small:2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o run -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/run.log 2>&1
find $OUT/pmc -name "*.db" -delete
cd $ROOT
python3 scripts/mfma_pmc_summary.py $OUT/pmc $WL | tee $OUT/mfma_pmc.md
