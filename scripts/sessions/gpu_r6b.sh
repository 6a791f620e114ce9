#!/bin/bash
# full GPU suite + the default bench line (three legs)      usage: gpu_r6b.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2700 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"])
print("e2e", d["end_to_end"]["time_solve"], d["end_to_end"]["setup_seconds"], "cpu", d["end_to_end"].get("cpu"))
for key in ("at_scale", "at_scale_large"):
    a=d[key]; f=a["roofline_families"]
    print(key, a["value"], a["ms_per_step"], {k: (round(v["ms_per_step"],2), round(v.get("achieved_tflops", v["frac"]),3)) for k,v in f.items()}, a["end_to_end"], a["model_build_seconds"])
print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value","cores","setup_seconds")}, d["cpu_baseline"]["end_to_end"])
PY
