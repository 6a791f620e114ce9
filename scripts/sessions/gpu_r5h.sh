#!/bin/bash
# round 4, session 5h: leaf size of the dissection / merged levels on big problems
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5h
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
    print("$wl $tag", "ms/step %.2f" % r["ms_per_step"], "steps/s %.2f" % r["value"], "factor %.2f ms" % f["factor"]["ms_per_step"],
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "GF %.0f nnz %.0fM levels %d fronts %d" % (s["factor_flops"]/1e9, s["factor_nnz"]/1e6, s["nr_level"], s["nr_front"]), flush=True)
except Exception as e:
    print("$wl $tag failed", e, open("$f.err").read()[-800:])
PY
}
for wl in refine:armadillo_small:1 block:48 refine:armadillo_small:2; do
  for leaf in 16 32 48 64 96; do run leaf$leaf $wl SANM_MF_LEAF=$leaf; done
  run leaf32_m1 $wl SANM_MF_MERGE=1
  run leaf32_m13 $wl SANM_MF_MERGE=1,3
done
