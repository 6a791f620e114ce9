#!/bin/bash
# A/B of two builds of the library on one box (SANM_HIP_LIBRARY)     usage: gpu_r6z.sh <tag> <libA> <libB>
set -u
TAG=$1; LA=$2; LB=$3
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
for rep in 1 2 3; do
  run x8_A_$rep refine:armadillo_small:1 10 SANM_HIP_LIBRARY=$ROOT/$LA
  run x8_B_$rep refine:armadillo_small:1 10 SANM_HIP_LIBRARY=$ROOT/$LB
done
for rep in 1 2; do
  run x64_A_$rep refine:armadillo_small:2 3 SANM_HIP_LIBRARY=$ROOT/$LA
  run x64_B_$rep refine:armadillo_small:2 3 SANM_HIP_LIBRARY=$ROOT/$LB
done
