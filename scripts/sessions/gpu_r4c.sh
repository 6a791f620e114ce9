#!/bin/bash
# round 4, session c: fused next_coeff A/B on one box, tail trace, parity tests, arbiter, leaf x merge sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_device_anm.py tests/test_device_ops.py tests/test_gpu_fullsize.py tests/test_fault_injection.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
# bit-identity of the variants (md5 of x_1, x_2, x_8, x_N, accepted range of the first expansion)
for c in human_arap16 armadillo_small; do
  SANM_ORDER1_HOST=1 SANM_NO_NEXT_COEFF_FUSION=1 python scripts/determinism.py $c --tag r3_path 2>/dev/null | tee -a $OUT/determinism.jsonl
  SANM_ORDER1_HOST=1 python scripts/determinism.py $c --tag fused_next_coeff 2>/dev/null | tee -a $OUT/determinism.jsonl
  python scripts/determinism.py $c --tag fused_and_device_order1 2>/dev/null | tee -a $OUT/determinism.jsonl
done
for rep in 1 2; do
  SANM_ORDER1_HOST=1 SANM_NO_NEXT_COEFF_FUSION=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_unfused_$rep.json 2>/dev/null
  SANM_ORDER1_HOST=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_fused_$rep.json 2>/dev/null
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_both_$rep.json 2>/dev/null
done
python - <<PY
import json
for k in ("unfused_1","fused_1","both_1","unfused_2","fused_2","both_2"):
    d=json.load(open("$OUT/bench_%s.json"%k)); f=d["roofline_families"]
    print(k, "%.1f steps/s %.3f ms"%(d["value"],d["ms_per_step"]), {n:(round(v["ms_per_step"],3), v["launches_per_step"]) for n,v in f.items()})
PY
SANM_TAIL_TRACE=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/tail_trace.txt
grep tail_trace $OUT/tail_trace.txt
timeout 1500 python scripts/pade_arbiter.py --api hip --out $OUT/r04_pade_arbiter.json > $OUT/arbiter.log 2>&1
tail -3 $OUT/arbiter.log
bash scripts/sweep_leaf_merge.sh r4c_sweep > /dev/null 2>&1
cat $ROOT/gpurun_out/r4c_sweep/sweep.md | head -60
