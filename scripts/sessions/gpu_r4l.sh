#!/bin/bash
# round 4, session l: defaults of the chains / wide backward kernel confirmed on block:32/48/60 and the small meshes; trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4l
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
run default block:48 X=1
run default block:32 X=1
run old block:32 SANM_MF_SPLIT_K=0 SANM_MF_WIDE_MIN_M=1000000
run default armadillo_small X=1
run default block:60 X=1
run old block:60 SANM_MF_SPLIT_K=0 SANM_MF_WIDE_MIN_M=1000000
run W768 block:60 SANM_MF_SPLIT_K=768
run W1536 block:60 SANM_MF_SPLIT_K=1536
