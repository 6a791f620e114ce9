#!/bin/bash
# small-front kernel: the direct-solver GPU tests, then factor times of the two bigger legs      usage: gpu_r6_sf.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_direct_solver.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for WL in refine:armadillo_small:1 refine:armadillo_small:2; do
  N=$(echo $WL | tr ':' '_')
  ST=10; [ $WL = refine:armadillo_small:2 ] && ST=3
  timeout 900 python bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$N.json").read().strip().splitlines()[-1])
f=d["roofline_families"]; e=d.get("end_to_end") or {}
print("$WL", round(d["value"],3), "steps/s", round(d["ms_per_step"],2), "ms", {k:(round(v["ms_per_step"],2), round(v.get("frac",0),3), round(v.get("achieved_tflops",0),2)) for k,v in f.items() if k in ("solve","factor","taylor")}, "e2e", round(e.get("time_solve",0),3), e.get("iter"))
PY
done
