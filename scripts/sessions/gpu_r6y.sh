#!/bin/bash
# small-front LU with one barrier per step: tests + timing     usage: gpu_r6y.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_direct_solver.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -2
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3))
PY
}
for rep in 1 2 3; do run x8_$rep refine:armadillo_small:1 10 A=1; done
for rep in 1 2; do run x64_$rep refine:armadillo_small:2 3 A=1; done
python scripts/determinism.py 2>/dev/null | tail -3
