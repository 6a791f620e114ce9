#!/bin/bash
# round 4, session f: tall tiles for the four triangular products of big fronts: bit identity and time against the build
# without them (-DSANM_MF_NO_TALL12), own zero fill against the runtime's
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4f
mkdir -p $OUT
cd $ROOT
: > $OUT/determinism.jsonl
for w in block:32 armadillo_small; do
SANM_MF_RUNTIME_FILL=1 SANM_HIP_LIBRARY=$ROOT/sanm_amd/libsanm_hip_notall12.so python scripts/determinism.py $w --tag round3_gemms_runtime_fill 2>&1 | tail -1 | tee -a $OUT/determinism.jsonl
python scripts/determinism.py $w --tag tall12_own_fill 2>&1 | tail -1 | tee -a $OUT/determinism.jsonl
done
bash scripts/gpu_ab_lib.sh "notall12 default notall12 default" "block:32 block:48"
SANM_MF_RUNTIME_FILL=1 timeout 600 python bench.py --steps 6 --warmup 2 --workload block:48 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline_families']; print('runtime fill block:48', round(d['ms_per_step'],2), 'factor', round(f['factor']['ms_per_step'],2))"
timeout 900 python -m pytest tests/test_direct_solver.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
