#!/bin/bash
# round 5: environment variants of the at-scale workload on one box (family times of the bench line)
# usage: bash scripts/gpu_r5_ab.sh <tag> <workload> "VAR=val VAR2=val" "..." ...
set -u
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
i=0
for v in "base" "$@"; do
  i=$((i+1))
  if [ "$v" = "base" ]; then envs=""; else envs="$v"; fi
  env $envs timeout 600 python bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/b_$i.json 2> $OUT/b_$i.err
  python - "$OUT/b_$i.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    f = d["roofline_families"]
    print("%-44s %7.2f steps/s  step %6.2f ms | factor %6.2f (%5.1f TF/s) solve %6.2f (%.3f) taylor %5.2f | levels %d fronts %d nnz %.1fM" % (
        sys.argv[2], d["value"], d["ms_per_step"], f["factor"]["ms_per_step"], f["factor"]["achieved_tflops"],
        f["solve"]["ms_per_step"], f["solve"]["frac"], f["taylor"]["ms_per_step"], d["config"]["solver_stats"]["nr_level"],
        d["config"]["solver_stats"]["nr_front"], d["config"]["solver_stats"]["factor_nnz"] / 1e6), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
done
