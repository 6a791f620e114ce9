#!/bin/bash
# tall-tile lists + A/B of the zero-fill / assigned Schur blocks on one box     usage: gpu_r6e.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_direct_solver.py tests/test_gpu_dist.py -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("$name", round(d["value"],3), round(d["ms_per_step"],2), {k: round(v["ms_per_step"],2) for k,v in f.items()}, "factor TF", round(f["factor"]["achieved_tflops"],2))
PY
}
for rep in 1 2; do
  run x8_new_$rep refine:armadillo_small:1 10 A=1
  run x8_nola_$rep refine:armadillo_small:1 10 SANM_MF_NO_LOOKAHEAD=1
  run x8_la5_$rep refine:armadillo_small:1 10 SANM_MF_LOOKAHEAD_MIN_GF=5
  run small_nola_$rep armadillo_small 20 SANM_MF_NO_LOOKAHEAD=1
  run small_new_$rep armadillo_small 20 A=1
done
run x64_new refine:armadillo_small:2 3 A=1
run x64_nola refine:armadillo_small:2 3 SANM_MF_NO_LOOKAHEAD=1
run b48_new block:48 3 A=1
run b48_nola block:48 3 SANM_MF_NO_LOOKAHEAD=1
run x64_la5 refine:armadillo_small:2 3 SANM_MF_LOOKAHEAD_MIN_GF=5
run x64_la60 refine:armadillo_small:2 3 SANM_MF_LOOKAHEAD_MIN_GF=60
