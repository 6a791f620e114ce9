#!/bin/bash
# round 4, session k: the backward kernel for long rows (bwd_wide_kernel): forced on small fronts in the solver tests,
# rows per workgroup and the split width on block:48 / block:32
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4k
mkdir -p $OUT
cd $ROOT
for R in 1 2 4; do
SANM_MF_WIDE_MIN_M=0 SANM_MF_WIDE_R=$R SANM_MF_SPLIT_K=48 timeout 900 python -m pytest tests/test_direct_solver.py -x -q -m gpu > $OUT/pytest_wide_R$R.log 2>&1; tail -2 $OUT/pytest_wide_R$R.log
done
run() {  # tag, workload, env...
  local tag=$1 wl=$2; shift 2
  env "$@" timeout 900 python bench.py --steps 4 --warmup 2 --workload $wl --no-cpu-baseline > $OUT/bench_${wl/:/}_$tag.json 2> $OUT/bench_${wl/:/}_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_${wl/:/}_$tag.json").read().strip().splitlines()[-1])
    f = r["roofline_families"]; s = r["config"]["solver_stats"]
    print("$wl $tag", "ms/step %.2f" % r["ms_per_step"], "factor %.2f ms %.1f TF" % (f["factor"]["ms_per_step"], f["factor"]["achieved_tflops"]),
          "solve %.2f ms frac %.3f" % (f["solve"]["ms_per_step"], f["solve"]["frac"]), "GF %.0f levels %d" % (s["factor_flops"] / 1e9, s["nr_level"]), flush=True)
except Exception as e:
    print("$wl $tag failed", e)
PY
}
run nowide block:48 SANM_MF_WIDE_MIN_M=1000000
run wide_auto block:48 X=1
run wide_R1 block:48 SANM_MF_WIDE_R=1
run wide_R2 block:48 SANM_MF_WIDE_R=2
run wide_R4 block:48 SANM_MF_WIDE_R=4
run wide_m1024 block:48 SANM_MF_WIDE_MIN_M=1024
run wide_m4096 block:48 SANM_MF_WIDE_MIN_M=4096
run W768 block:48 SANM_MF_SPLIT_K=768
run W1280 block:48 SANM_MF_SPLIT_K=1280
run W1536 block:48 SANM_MF_SPLIT_K=1536
run nowide block:32 SANM_MF_WIDE_MIN_M=1000000
run wide_auto block:32 X=1
run wide_m1024 block:32 SANM_MF_WIDE_MIN_M=1024
run W768 block:32 SANM_MF_SPLIT_K=768
run W1536 block:32 SANM_MF_SPLIT_K=1536
