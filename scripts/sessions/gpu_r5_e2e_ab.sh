#!/bin/bash
# end_to_end of a workload with an environment switch off / on, three times each     usage: gpu_r5_e2e_ab.sh <tag> <VAR> <VAL> <workload>
set -u
TAG=$1; VAR=$2; VAL=$3; WL=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for K in off on off on off on; do
  if [ $K = on ]; then export $VAR=$VAL; else unset $VAR; fi
  timeout 300 python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/e2e_$K.json 2> $OUT/e2e_$K.err
  python - <<PY
import json
d=json.loads(open("$OUT/e2e_$K.json").read().strip().splitlines()[-1])
e=d["end_to_end"]
print("$WL", "$VAR", "$K", "cold", round(e["cold"]["time_solve"],4), "ctor", round(e["cold"]["constructor_seconds"],4), "cached", round(e["cached"]["time_solve"],4), "ctor", round(e["cached"].get("constructor_seconds",0),4), e["cached"]["setup_seconds"])
PY
done
