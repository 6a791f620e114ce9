#!/bin/bash
# constructor laps (SANM_DEBUG_SETUP, SANM_MF_DEBUG) and the end-to-end figure of the bigger legs   usage: gpu_r6_setup.sh <tag> [workloads...]
set -u
TAG=$1; shift
WLS=${*:-armadillo_small refine:armadillo_small:1 refine:armadillo_small:2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
nproc
for WL in $WLS; do
  N=$(echo $WL | tr ':' '_')
  ST=3; [ $WL = refine:armadillo_small:2 ] && ST=2
  SANM_DEBUG_SETUP=1 SANM_MF_DEBUG=1 timeout 900 python bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  echo "== $WL"
  grep "^\[setup\]\|mf analysis" $OUT/bench_$N.err | grep -v "mf level" | awk '{a[$0]++; if (a[$0]==1) print}' | tail -60
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$N.json").read().strip().splitlines()[-1])
e=d["end_to_end"]
print("$WL", "value", round(d["value"],2), "e2e", round(e["time_solve"],4), e["iter"], "ctor", e.get("constructor_seconds"), {k:(round(v,4) if isinstance(v,float) else v) for k,v in e["setup_seconds"].items()})
PY
done
