#!/bin/bash
# round 4, session 5c: organic meshes at scale: armadillo_small with every tet cut into 8 (338 k tets) and 64 (2.7 M)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c
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
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "taylor %.2f" % f["taylor"]["ms_per_step"], "GF %.0f levels %d nnz %.0fM" % (s["factor_flops"] / 1e9, s["nr_level"], s["factor_nnz"]/1e6), r["config"]["workload"][:90], flush=True)
except Exception as e:
    print("$wl $tag failed", e, open("$f.err").read()[-800:])
PY
}
run default refine:armadillo_small:1 X=1
run noaxis refine:armadillo_small:1 SANM_MF_AXIS_CUTS_MIN=0
run default refine:human_arap16:1 X=1
run default refine:armadillo_small:2 X=1
