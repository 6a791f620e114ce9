#!/bin/bash
# look-ahead on / off alternating on one box, 2.7 M tets and block:48      usage: gpu_r6i.sh <tag>
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
print("$name", round(d["value"],3), round(d["ms_per_step"],2), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],2))
PY
}
for rep in 1 2 3 4; do
  run x64_la_$rep refine:armadillo_small:2 5 A=1
  run x64_nola_$rep refine:armadillo_small:2 5 SANM_MF_NO_LOOKAHEAD=1
done
for rep in 1 2 3; do
  run b48_la_$rep block:48 5 A=1
  run b48_nola_$rep block:48 5 SANM_MF_NO_LOOKAHEAD=1
done
