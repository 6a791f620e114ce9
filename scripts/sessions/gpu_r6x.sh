#!/bin/bash
# old switches on the last tree, headline mesh: graph replay of the sweeps, merged top block     usage: gpu_r6x.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
    print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],3), "solve", round(f["solve"]["ms_per_step"],3), "tail", round(f["tail"]["ms_per_step"],3))
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/$name.err").read()[-500:])
PY
}
for rep in 1 2; do
  run small_default_$rep armadillo_small 40 A=1
  run small_graph_$rep armadillo_small 40 SANM_MF_GRAPH=1
  run small_top_$rep armadillo_small 40 SANM_MF_TOP=2048
  run small_top_graph_$rep armadillo_small 40 SANM_MF_TOP=2048 SANM_MF_GRAPH=1
done
