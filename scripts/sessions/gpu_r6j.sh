#!/bin/bash
# solve kernels: preloaded chunks by the typical row instead of the longest, alternating on one box     usage: gpu_r6j.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3))
PY
}
timeout 900 python -m pytest tests/test_direct_solver.py -q -m gpu -x 2>&1 | tail -2
SANM_MF_LS_WIDTH=1.25 timeout 900 python -m pytest tests/test_direct_solver.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do
  for w in 0 1.0 1.25 1.6; do
    run x8_w${w}_$rep refine:armadillo_small:1 10 SANM_MF_LS_WIDTH=$w
  done
done
for rep in 1 2; do
  for w in 0 1.25; do
    run small_w${w}_$rep armadillo_small 20 SANM_MF_LS_WIDTH=$w
    run x64_w${w}_$rep refine:armadillo_small:2 3 SANM_MF_LS_WIDTH=$w
  done
done
