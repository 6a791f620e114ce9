#!/bin/bash
# per-level trace of the solves at 338 k tets with the new chunk selection + more factors   usage: gpu_r6k.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash scripts/sessions/gpu_r5_trace.sh $TAG/trace_x8 refine:armadillo_small:1 3 > /dev/null 2>&1
grep -A16 "== fwd_level_tr\|== fwd_level_kernel\|== bwd_level" gpurun_out/$TAG/trace_x8/by_grid.txt | head -70
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3))
PY
}
for rep in 1 2; do
  for w in 1.25 0.75 0.5; do
    run x8_w${w}_$rep refine:armadillo_small:1 10 SANM_MF_LS_WIDTH=$w
  done
  run x8_r2_$rep refine:armadillo_small:1 10 SANM_MF_LS_R=2
  run x8_fsr1_$rep refine:armadillo_small:1 10 SANM_MF_FS_R=1
  run x8_fsr4_$rep refine:armadillo_small:1 10 SANM_MF_FS_R=4
done
