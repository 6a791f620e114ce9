#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3e
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_direct_solver.py tests/test_device_anm.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python scripts/determinism.py armadillo_small --tag newgemm > $OUT/det.log 2>&1; grep "^{" $OUT/det.log
for w in armadillo_small block:32 block:48; do
  timeout 900 python bench.py --steps 6 --warmup 2 --workload $w --no-cpu-baseline > $OUT/bench_${w/:/}.json 2> $OUT/bench_${w/:/}.err; echo "bench $w rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/bench_${w/:/}.json"))
f=d["roofline_families"]
print("$w", round(d["value"],2), "steps/s", round(d["ms_per_step"],3), "ms; factor", round(f["factor"]["ms_per_step"],2), "ms", round(f["factor"]["achieved_tflops"],1), "TF; solve", round(f["solve"]["ms_per_step"],2), "ms frac", round(f["solve"]["frac"],3))
PY
done
