#!/bin/bash
# A/B on one box: transposed forward operator up to 128 / 256 pivots     usage: gpu_r5_fwdt.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_direct_solver.py -q -m gpu -x 2>&1 | tail -2
for WL in refine:armadillo_small:1 armadillo_small; do
  N=$(echo $WL | tr ':' '_')
  for K in 128 256 128 256; do
    SANM_MF_FWD_T_MAX_K=$K timeout 300 python bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/${N}_$K.json 2> $OUT/${N}_$K.err
    python - <<PY
import json
d=json.loads(open("$OUT/${N}_$K.json").read().strip().splitlines()[-1])
f=d["roofline_families"]
print("$WL", $K, "ms/step", round(d["ms_per_step"],3), "solve", round(f["solve"]["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],3))
PY
  done
done
