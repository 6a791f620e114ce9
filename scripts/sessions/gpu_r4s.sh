#!/bin/bash
# round 4, session s: look-ahead of the factorisation: solver tests (bit-identity on / off), A/B at scale, md5s
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4s
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_direct_solver.py -x -q -m gpu > $OUT/pytest_solver.log 2>&1; tail -3 $OUT/pytest_solver.log
for la in 0 1; do
  SANM_MF_LOOKAHEAD=$la timeout 600 python scripts/determinism.py block:32 --tag lookahead_$la >> $OUT/determinism.jsonl 2>> $OUT/determinism.err
done
cut -c1-200 $OUT/determinism.jsonl
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
    print("$wl $tag failed", e, open("$OUT/bench_${wl/:/}_$tag.err").read()[-600:])
PY
}
for wl in block:48 block:32 block:60 block:40; do
  run la0 $wl SANM_MF_LOOKAHEAD=0
  run la1 $wl X=1
done
