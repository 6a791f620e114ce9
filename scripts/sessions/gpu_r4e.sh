#!/bin/bash
# round 4, session e: sanity checks behind a mark A/B, parity tests that touch the reordered tail
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4e
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_device_anm.py tests/test_fault_injection.py tests/test_gpu_fullsize.py tests/test_vector_graphs.py tests/test_tikhonov.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for c in human_arap16 armadillo_small; do
  SANM_SANITY_FIRST=1 python scripts/determinism.py $c --tag sanity_first 2>/dev/null | tee -a $OUT/determinism.jsonl
  python scripts/determinism.py $c --tag sanity_behind 2>/dev/null | tee -a $OUT/determinism.jsonl
done
for rep in 1 2 3; do
  SANM_SANITY_FIRST=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_first_$rep.json 2>/dev/null
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_behind_$rep.json 2>/dev/null
done
python - <<PY
import json
for k in ("first_1","behind_1","first_2","behind_2","first_3","behind_3"):
    d=json.load(open("$OUT/bench_%s.json"%k)); f=d["roofline_families"]
    print(k, "%.1f steps/s %.3f ms"%(d["value"],d["ms_per_step"]), {n:(round(v["ms_per_step"],3), v["launches_per_step"]) for n,v in f.items()})
PY
SANM_TAIL_TRACE=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/tail_trace.txt
grep tail_trace $OUT/tail_trace.txt
