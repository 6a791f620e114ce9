#!/bin/bash
# the other BASELINE configs and the block sizes on the last tree     usage: gpu_r5_others.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for WL in bob human_arap16 block:32 block:48 refine:armadillo_small:2; do
  N=$(echo $WL | tr ':' '_')
  ST=8; [ $WL = block:48 ] && ST=3; [ $WL = refine:armadillo_small:2 ] && ST=2
  timeout 900 python bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$N.json").read().strip().splitlines()[-1])
    f=d["roofline_families"]; e=d.get("end_to_end") or {}
    print("$WL", round(d["value"],2), "steps/s", round(d["ms_per_step"],2), "ms", {k:(round(v["ms_per_step"],2), round(v.get("frac",0),3), round(v.get("achieved_tflops",0),1)) for k,v in f.items() if k in ("solve","factor","taylor")}, "e2e", round(e.get("time_solve",0),3), e.get("iter"))
except Exception as ex:
    print("$WL FAILED", ex)
PY
done
