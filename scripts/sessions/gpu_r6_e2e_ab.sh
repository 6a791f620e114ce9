#!/bin/bash
# end-to-end figure (second solve: cached kernels) of a leg under alternating settings on one box
# usage: gpu_r6_e2e_ab.sh <tag> <workload> <reps> <name=ENV=VAL[,ENV=VAL]>...
set -u
TAG=$1; WL=$2; REPS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $(echo $envs | tr ',' ' ' | sed "s#ROOT#$ROOT#g") timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/${name}_$rep.json 2> $OUT/${name}_$rep.err
    python - <<PY
import json
d=json.loads(open("$OUT/${name}_$rep.json").read().strip().splitlines()[-1])
e=d["end_to_end"]; c=e["cached"]["setup_seconds"]
print("$name $rep", "e2e cold", round(e["cold"]["time_solve"],4), "cached", round(e["cached"]["time_solve"],4), "thread", c.get("analysis_thread"), "wait", c.get("analysis"), "tables", round(sum(c.get(k,0) for k in ("tet_order","pattern","program","remap_tables")),4), "device", c.get("analysis_device"))
PY
  done
done
