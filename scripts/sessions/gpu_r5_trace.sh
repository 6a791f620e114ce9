#!/bin/bash
# kernel trace + stats of one workload (per-launch-shape tables: scripts/prof_by_grid.py)   usage: gpu_r5_trace.sh <tag> <workload> [steps]
set -u
TAG=$1; WL=$2; ST=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/stats.log 2>&1
cd $ROOT
python scripts/prof_summary.py $OUT/stats > $OUT/kernel_stats.md 2>/dev/null
for k in fwd_level_tr fwd_level_kernel bwd_level small_front update_kernel gemm2_list gemm1_list extend_add; do echo "== $k"; python scripts/prof_by_grid.py $OUT/stats $k | head -30; done > $OUT/by_grid.txt 2>&1
head -40 $OUT/kernel_stats.md
