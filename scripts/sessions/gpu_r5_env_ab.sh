#!/bin/bash
# A/B of one environment switch on one box: bench legs with VAR unset / VAR=VAL, twice each      usage: gpu_r5_env_ab.sh <tag> <VAR> <VAL> [workloads...]
set -u
TAG=$1; VAR=$2; VAL=$3; shift 3
WLS=${@:-"refine:armadillo_small:1 armadillo_small"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for WL in $WLS; do
  N=$(echo $WL | tr ':' '_')
  for K in off on off on; do
    if [ $K = on ]; then export $VAR=$VAL; else unset $VAR; fi
    timeout 300 python bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/${N}_$K.json 2> $OUT/${N}_$K.err
    python - <<PY
import json
d=json.loads(open("$OUT/${N}_$K.json").read().strip().splitlines()[-1])
f=d["roofline_families"]
print("$WL", "$VAR", "$K", "ms/step", round(d["ms_per_step"],3), "solve", round(f["solve"]["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],3))
PY
  done
done
