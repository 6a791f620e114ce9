#!/bin/bash
# phase clock of small_front_kernel for variant builds (-DSANM_SF_PHASES) shipped as sanm_amd/variant_<name>.so    usage: gpu_r6_sfph.sh <tag> <names...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for V in "$@"; do
for WL in refine:armadillo_small:1; do
  N=$(echo $WL | tr ':' '_')
  SANM_HIP_LIBRARY=$ROOT/sanm_amd/variant_$V.so timeout 600 python bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/ph_${V}_$N.json 2> $OUT/ph_${V}_$N.err
  echo "== $V $WL"; grep "small_front_kernel, per" $OUT/ph_${V}_$N.err | tail -1
  python - <<PY
import json
d=json.loads(open("$OUT/ph_${V}_$N.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("   factor", round(f["factor"]["ms_per_step"],2))
PY
done
done
