#!/bin/bash
# per-level trace of the solves at 2.7 M tets      usage: gpu_r6n.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash scripts/sessions/gpu_r5_trace.sh $TAG/trace_x64 refine:armadillo_small:2 2 > /dev/null 2>&1
for k in fwd_level_tr fwd_level_kernel bwd_level bwd_wide; do echo "== $k"; python scripts/prof_by_grid.py gpurun_out/$TAG/trace_x64/stats $k | head -60; done
SANM_MF_DEBUG=1 timeout 300 python bench.py --workload refine:armadillo_small:2 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none 2>&1 | grep "mf level" | head -30
