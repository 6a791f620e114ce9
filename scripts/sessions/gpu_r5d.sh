#!/bin/bash
# round 4, session 5d: cut positions 40 ... 60 % in the dissection of big problems, on the refined organic meshes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5d
mkdir -p $OUT
cd $ROOT
run() {  # tag, workload, env...
  local tag=$1 wl=$2; shift 2
  local f=$OUT/bench_$(echo $wl | tr ':' '_')_$tag
  env "$@" timeout 2400 python bench.py --steps 5 --warmup 2 --workload $wl --no-cpu-baseline > $f.json 2> $f.err
  python - <<PY
import json
try:
    r = json.loads(open("$f.json").read().strip().splitlines()[-1])
    f = r["roofline_families"]; s = r["config"]["solver_stats"]
    print("$wl $tag", "ms/step %.2f" % r["ms_per_step"], "steps/s %.2f" % r["value"], "factor %.2f ms %.1f TF" % (f["factor"]["ms_per_step"], f["factor"]["achieved_tflops"]),
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "GF %.0f levels %d nnz %.0fM" % (s["factor_flops"] / 1e9, s["nr_level"], s["factor_nnz"]/1e6), flush=True)
except Exception as e:
    print("$wl $tag failed", e, open("$f.err").read()[-800:])
PY
}
for wl in refine:armadillo_small:1 refine:human_arap16:1 refine:bob:1 refine:armadillo_small:2 block:48; do
  run std $wl X=1
  run wide $wl SANM_MF_ND_WIDE=1
done
