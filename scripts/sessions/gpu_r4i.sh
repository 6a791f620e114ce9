#!/bin/bash
# round 4, session i: chains in place of big fronts (multifrontal.cpp, split_big_fronts): the split width W swept on
# the block workloads on one box, the solver tests with the path forced on small fronts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4i
mkdir -p $OUT
cd $ROOT
SANM_MF_SPLIT_K=48 timeout 900 python -m pytest tests/test_direct_solver.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest_split48.log 2>&1; tail -3 $OUT/pytest_split48.log
for wl in block:48 block:32; do
  for W in 0 2048 1024 768 512; do
    SANM_MF_SPLIT_K=$W timeout 900 python bench.py --steps 4 --warmup 2 --workload $wl --no-cpu-baseline > $OUT/bench_${wl/:/}_W$W.json 2> $OUT/bench_${wl/:/}_W$W.err
    python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_${wl/:/}_W$W.json").read().strip().splitlines()[-1])
    f = r["roofline_families"]; s = r["config"]["solver_stats"]
    print("$wl W=$W", "ms/step %.2f" % r["ms_per_step"], "factor %.2f ms %.1f TF" % (f["factor"]["ms_per_step"], f["factor"]["achieved_tflops"]),
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "GF %.0f levels %d" % (s["factor_flops"] / 1e9, s["nr_level"]), flush=True)
except Exception as e:
    print("$wl W=$W failed", e)
PY
  done
done
