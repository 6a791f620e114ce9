#!/bin/bash
# round 4, session 5g: packed forward operators: solver tests, A/B at scale (blocks and refined organic meshes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5g
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_direct_solver.py -x -q -m gpu > $OUT/pytest_solver.log 2>&1; tail -3 $OUT/pytest_solver.log
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
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "nnz %.0fM" % (s["factor_nnz"]/1e6), flush=True)
except Exception as e:
    print("$wl $tag failed", e, open("$f.err").read()[-800:])
PY
}
for wl in refine:armadillo_small:1 refine:armadillo_small:2 block:32 block:48 block:60 refine:human_arap16:1; do
  run nopack $wl SANM_MF_PACK_FWD=0
  run pack $wl SANM_MF_PACK_FWD=1
done
