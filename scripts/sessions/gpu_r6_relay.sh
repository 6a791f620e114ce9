#!/bin/bash
# relay launches: the direct-solver GPU tests, then the three legs with and without (SANM_MF_RELAY_MAX_FRONTS=0), alternating    usage: gpu_r6_relay.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_direct_solver.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3), round(f["solve"]["frac"],3))
PY
}
for rep in 1 2; do
  run small_relay_$rep armadillo_small 20 SANM_X=1
  run small_levels_$rep armadillo_small 20 SANM_MF_RELAY_MAX_FRONTS=0
  run x8_relay_$rep refine:armadillo_small:1 10 SANM_X=1
  run x8_levels_$rep refine:armadillo_small:1 10 SANM_MF_RELAY_MAX_FRONTS=0
done
run x64_relay refine:armadillo_small:2 3 SANM_X=1
run x64_levels refine:armadillo_small:2 3 SANM_MF_RELAY_MAX_FRONTS=0
