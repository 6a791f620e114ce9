#!/bin/bash
# round 4, session m: chunk width of the chains from the cost model (alpha sqrt(b + k/2)) against fixed widths
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4m
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
for wl in block:48 block:60 block:32; do
  for A in 10 12 14 16 20; do run A$A $wl SANM_MF_SPLIT_ALPHA=$A; done
  run W1024 $wl SANM_MF_SPLIT_K=1024
  run W1536 $wl SANM_MF_SPLIT_K=1536
done
