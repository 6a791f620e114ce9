#!/bin/bash
# round-end session: full GPU suite, the default bench line, profiles (kernel stats + PMC of the headline and the 338 k-tet
# leg, kernel stats of the 2.7 M-tet leg), the other workloads      usage: gpu_r6_final.sh <tag> [parts: tests bench prof others]
set -u
TAG=$1; shift
PARTS=${*:-tests bench prof others}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  timeout 2700 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
  cp gpurun_out/parity_steps_*.json $OUT/ 2>/dev/null
fi
if has bench; then
  ( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], {k: round(v["ms_per_step"],3) for k,v in d["roofline_families"].items()})
print("e2e", d["end_to_end"]["time_solve"], d["end_to_end"]["setup_seconds"])
for key in ("at_scale", "at_scale_large"):
    a=d[key]; f=a["roofline_families"]
    print(key, a["value"], a["ms_per_step"], {k: (round(v["ms_per_step"],2), round(v.get("achieved_tflops", v["frac"]),3)) for k,v in f.items()}, a["end_to_end"]["time_solve"], a["end_to_end"]["iter"], a["end_to_end"]["setup_seconds"])
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["end_to_end"])
PY
fi
if has prof; then
  bash scripts/collect_profiles.sh ${TAG} armadillo_small 12 4 > $OUT/prof_small.log 2>&1
  bash scripts/collect_profiles.sh ${TAG}_x8 refine:armadillo_small:1 6 2 > $OUT/prof_x8.log 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_x64/stats -o run -- python3 $ROOT/bench.py --workload refine:armadillo_small:2 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/prof_x64.log 2>&1
  find $ROOT/gpurun_out/prof_${TAG}_x64 -name "*.db" -delete
  cd $ROOT
  ls gpurun_out/prof_${TAG}* | head -30
fi
if has others; then
  for WL in bob human_arap16 block:32 block:48; do
    N=$(echo $WL | tr ':' '_')
    ST=8; [ $WL = block:48 ] && ST=3
    timeout 900 python bench.py --workload $WL --steps $ST --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/bench_$N.json 2> $OUT/bench_$N.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$N.json").read().strip().splitlines()[-1])
    f=d["roofline_families"]; e=d.get("end_to_end") or {}
    print("$WL", round(d["value"],2), "steps/s", round(d["ms_per_step"],2), "ms", {k:(round(v["ms_per_step"],2), round(v.get("frac",0),3), round(v.get("achieved_tflops",0),1)) for k,v in f.items() if k in ("solve","factor","taylor")}, "e2e", round(e.get("time_solve",0),3), e.get("iter"))
except Exception as ex:
    print("$WL FAILED", ex)
PY
  done
fi
