#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3d
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_direct_solver.py tests/test_device_anm.py tests/test_fault_injection.py tests/test_tikhonov.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python scripts/determinism.py armadillo_small --tag graph > $OUT/det.log 2>&1
SANM_NO_MF_GRAPH=1 python scripts/determinism.py armadillo_small --tag nograph >> $OUT/det.log 2>&1
grep "^{" $OUT/det.log
for v in 0 1; do
  if [ $v = 1 ]; then export SANM_NO_MF_GRAPH=1; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "bench rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$v.json"))
print("NO_MF_GRAPH=$v", round(d["value"],1), "steps/s", round(d["ms_per_step"],3), "ms", {k:(round(f["ms_per_step"],3), f["launches_per_step"]) for k,f in d["roofline_families"].items()})
PY
done
