#!/bin/bash
# runtime environment switches of the HIP / HSA runtimes on the two bench legs     usage: gpu_r5_envs.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() {
  local name=$1; shift
  for WL in armadillo_small refine:armadillo_small:1; do
    N=$(echo $WL | tr ':' '_')
    env "$@" timeout 300 python bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/${N}_$name.json 2> $OUT/${N}_$name.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/${N}_$name.json").read().strip().splitlines()[-1])
    f=d["roofline_families"]
    print("$name", "$WL", "ms/step", round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in f.items()})
except Exception as e:
    print("$name", "$WL", "FAILED", e)
PY
  done
}
run base A=1
run devkernarg HIP_FORCE_DEV_KERNARG=1
run nointerrupt HSA_ENABLE_INTERRUPT=0
run both HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
run base2 A=1
