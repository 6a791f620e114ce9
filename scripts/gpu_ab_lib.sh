#!/bin/bash
# A/B of library builds (scripts/build_variants.py): bench lines per variant and workload
# usage: bash scripts/gpu_ab_lib.sh "<variant names, '' = default>" "<workloads>"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_lib
mkdir -p $OUT
cd $ROOT
for w in $2; do
 for v in $1; do
  if [ "$v" = "default" ]; then unset SANM_HIP_LIBRARY; else export SANM_HIP_LIBRARY=$ROOT/sanm_amd/libsanm_hip_$v.so; fi
  timeout 900 python bench.py --steps 6 --warmup 2 --workload $w --no-cpu-baseline > $OUT/bench_${v}_${w/:/}.json 2> $OUT/bench_${v}_${w/:/}.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${v}_${w/:/}.json"))
    f=d["roofline_families"]
    print("$v $w", round(d["value"],2), "steps/s", round(d["ms_per_step"],3), "ms; factor", round(f["factor"]["ms_per_step"],2), "ms", round(f["factor"]["achieved_tflops"],1), "TF; solve", round(f["solve"]["ms_per_step"],2))
except Exception as e:
    print("$v $w failed", e)
PY
 done
done
