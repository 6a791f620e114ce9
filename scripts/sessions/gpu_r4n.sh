#!/bin/bash
# round 4, session n: fixed chunk widths, finer
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4n
mkdir -p $OUT
cd $ROOT
run() {  # tag, workload, env...
  local tag=$1 wl=$2; shift 2
  env "$@" timeout 1200 python bench.py --steps 4 --warmup 2 --workload $wl --no-cpu-baseline > $OUT/bench_${wl/:/}_$tag.json 2> $OUT/bench_${wl/:/}_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_${wl/:/}_$tag.json").read().strip().splitlines()[-1])
    f = r["roofline_families"]; s = r["config"]["solver_stats"]
    print("$wl $tag", "ms/step %.2f" % r["ms_per_step"], "factor %.2f ms %.1f TF" % (f["factor"]["ms_per_step"], f["factor"]["achieved_tflops"]),
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "GF %.0f levels %d" % (s["factor_flops"] / 1e9, s["nr_level"]), flush=True)
except Exception as e:
    print("$wl $tag failed", e)
PY
}
for W in 1280 1536 1792 2048 2560; do run W$W block:60 SANM_MF_SPLIT_K=$W; done
for W in 896 960 1024 1088 1152; do run W$W block:48 SANM_MF_SPLIT_K=$W; done
for W in 1024 1280 2048; do run W$W block:40 SANM_MF_SPLIT_K=$W; done
run W0 block:40 SANM_MF_SPLIT_K=0
