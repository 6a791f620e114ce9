#!/bin/bash
# selective zero-fill / assigned Schur blocks: solver + dist + full-size tests, the bench line, kernel stats of the 2.7 M-tet leg
# usage: gpu_r6d.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests/test_direct_solver.py tests/test_gpu_dist.py tests/test_device_anm.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
grep -h "4 ranks\|critical" $OUT/pytest_gpu.log | head -4
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], {k: round(v["ms_per_step"],3) for k,v in d["roofline_families"].items()})
for key in ("at_scale", "at_scale_large"):
    a=d[key]; f=a["roofline_families"]
    print(key, a["value"], a["ms_per_step"], {k: (round(v["ms_per_step"],2), round(v.get("achieved_tflops", v["frac"]),3)) for k,v in f.items()}, a["end_to_end"]["time_solve"])
PY
cd /tmp && export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x64 -o run -- python3 $ROOT/bench.py --workload refine:armadillo_small:2 --steps 2 --warmup 1 $FLAGS > $OUT/stats_x64.log 2>&1
find $OUT -name "*.db" -delete
cd $ROOT
python scripts/prof_summary.py $OUT/stats_x64 5 30 2>&1 | head -80
