#!/bin/bash
# two-phase threshold sweep at 2.7 M tets and block:48 (the sum over a height crosses 8e9 on levels of thousands of small fronts)
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
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3), "GF", round(d["config"]["solver_stats"]["factor_flops"]/1e9,1))
PY
}
for thr in 8e9 2e10 4e10 8e10 2e11; do
  run x64_tp$thr refine:armadillo_small:2 3 SANM_MF_TWO_PHASE=$thr
done
run x64_tp8e9_b refine:armadillo_small:2 3 SANM_MF_TWO_PHASE=8e9
for thr in 8e9 2e10 4e10 8e10; do
  run b48_tp$thr block:48 3 SANM_MF_TWO_PHASE=$thr
done
