#!/bin/bash
# round 4, session g: all extend-add rounds of a level in one launch: bit identity and time against the rounds
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4g
mkdir -p $OUT
cd $ROOT
: > $OUT/determinism.jsonl
for w in armadillo_small human_arap16 bob block:32; do
SANM_MF_EA_ROUNDS=1 python scripts/determinism.py $w --tag ea_rounds 2>&1 | tail -1 | tee -a $OUT/determinism.jsonl
python scripts/determinism.py $w --tag ea_one_launch 2>&1 | tail -1 | tee -a $OUT/determinism.jsonl
done
for rep in 1 2 3; do
  SANM_MF_EA_ROUNDS=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_rounds_$rep.json 2>/dev/null
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_one_$rep.json 2>/dev/null
done
python - <<PY
import json
for k in ("rounds_1","one_1","rounds_2","one_2","rounds_3","one_3"):
    d=json.load(open("$OUT/bench_%s.json"%k)); f=d["roofline_families"]
    print(k, "%.1f steps/s %.3f ms"%(d["value"],d["ms_per_step"]), {n:(round(v["ms_per_step"],3), v["launches_per_step"]) for n,v in f.items()})
PY
timeout 900 python -m pytest tests/test_direct_solver.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
