#!/bin/bash
# amalgamated heights (SANM_MF_MERGE) on the bigger legs, alternating on one box      usage: gpu_r6_merge.sh <tag>
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
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]; s=d.get("solver_stats") or d.get("stats") or {}
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3), round(f["solve"]["frac"],3), "nnz", s.get("factor_nnz"), "levels", s.get("nr_level"))
PY
}
for rep in 1 2; do
  run x8_none_$rep refine:armadillo_small:1 10 SANM_MF_MERGE=none
  run x8_1_$rep refine:armadillo_small:1 10 SANM_MF_MERGE=1
  run x8_13_$rep refine:armadillo_small:1 10 SANM_MF_MERGE=1,3
  run x8_135_$rep refine:armadillo_small:1 10 SANM_MF_MERGE=1,3,5
done
run x64_none refine:armadillo_small:2 3 SANM_MF_MERGE=none
run x64_1 refine:armadillo_small:2 3 SANM_MF_MERGE=1
run x64_13 refine:armadillo_small:2 3 SANM_MF_MERGE=1,3
run b48_none block:48 3 SANM_MF_MERGE=none
run b48_13 block:48 3 SANM_MF_MERGE=1,3
