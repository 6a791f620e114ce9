#!/bin/bash
# bench lines under environment variants (through gpurun): bash scripts/env_ab.sh <tag> "VAR=val" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
i=0
for v in "" "$@"; do
  for rep in 1 2; do
    if [ -n "$v" ]; then env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b_${i}_$rep.json 2>> $OUT/err.log
    else python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b_${i}_$rep.json 2>> $OUT/err.log; fi
    python - <<PY
import json
d=json.load(open("$OUT/b_${i}_$rep.json"))
f=d["roofline_families"]
print("variant $i [$v] rep $rep:", round(d["value"],2), "steps/s", round(d["ms_per_step"],3), "ms |", " ".join(f"{k} {f[k]['ms_per_step']:.3f}" for k in f))
PY
  done
  i=$((i+1))
done
