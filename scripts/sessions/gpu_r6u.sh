#!/bin/bash
# from how many fronts a level takes the small-front kernel     usage: gpu_r6u.sh <tag>
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
for rep in 1 2; do
  for m in 1024 512 256 128; do
    run x8_sf${m}_$rep refine:armadillo_small:1 10 SANM_MF_SMALL_MIN_FRONTS=$m
  done
done
for m in 1024 256; do
  run small_sf${m} armadillo_small 20 SANM_MF_SMALL_MIN_FRONTS=$m
  run x64_sf${m} refine:armadillo_small:2 3 SANM_MF_SMALL_MIN_FRONTS=$m
done
