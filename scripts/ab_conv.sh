#!/bin/bash
# A/B of the fused convolution loop of the compiled Taylor kernels (through gpurun): GPU tests of the operator
# level and the continuations, then bench lines with and without the fusion.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-ab_conv}
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_device_ops.py tests/test_device_anm.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for w in armadillo_small bob human_arap16; do
  for v in 0 1 2; do
    unset SANM_NO_CONV_FUSION SANM_NO_COEFF_OVERLAP
    if [ $v = 1 ]; then export SANM_NO_CONV_FUSION=1; fi
    if [ $v = 2 ]; then export SANM_NO_COEFF_OVERLAP=1; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --workload $w --no-cpu-baseline > $OUT/bench_${w}_nofuse$v.json 2>> $OUT/bench.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_${w}_nofuse$v.json"))
t=d["roofline_families"]["taylor"]
print("$w variant=$v (0 default, 1 no fusion, 2 no COEFF overlap)", round(d["value"],2), "steps/s", round(d["ms_per_step"],3), "ms | taylor ms", round(t["ms_per_step"],3), "frac", round(t["frac"],3))
PY
  done
done
