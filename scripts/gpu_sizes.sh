#!/bin/bash
# one bench line per synthetic block size on one box: how the step, the factorisation's rate and the solves' share of the
# HBM peak move with the size.  usage (through gpurun): bash scripts/gpu_sizes.sh <out dir under gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-sizes}
mkdir -p $OUT
cd $ROOT
echo "| workload | tets | unknowns | factor entries | GFLOP / factorisation | levels | ms / step | steps/s | factor ms | TFLOP/s | solve ms | of HBM peak | taylor ms |" > $OUT/sizes.md
echo "|---|---|---|---|---|---|---|---|---|---|---|---|---|" >> $OUT/sizes.md
for n in 12 16 24 32 40 48 56 60; do
  timeout 1500 python bench.py --steps 5 --warmup 2 --workload block:$n --no-cpu-baseline > $OUT/bench_block$n.json 2> $OUT/bench_block$n.err
  python - <<PY >> $OUT/sizes.md
import json, re
try:
    r = json.loads(open("$OUT/bench_block$n.json").read().strip().splitlines()[-1])
    f = r["roofline_families"]; s = r["config"]["solver_stats"]; w = r["config"]["workload"]
    T = re.search(r"T=(\d+)", w).group(1); N = re.search(r"n=(\d+)", w).group(1)
    print(f"| block:$n | {T} | {N} | {s['factor_nnz']/1e6:.1f} M | {s['factor_flops']/1e9:.1f} | {s['nr_level']} | {r['ms_per_step']:.2f} | {r['value']:.2f} | {f['factor']['ms_per_step']:.2f} | {f['factor']['achieved_tflops']:.1f} | {f['solve']['ms_per_step']:.2f} | {f['solve']['frac']:.3f} | {f['taylor']['ms_per_step']:.2f} |")
except Exception as e:
    print("| block:$n | failed:", e, "|")
PY
done
cat $OUT/sizes.md
