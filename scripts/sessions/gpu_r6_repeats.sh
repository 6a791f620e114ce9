#!/bin/bash
# the default bench line (without the CPU leg) several times on ONE box: the spread of the three legs      usage: gpu_r6_repeats.sh <tag> [n]
set -u
TAG=$1; N=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for i in $(seq 1 $N); do
  timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_$i.json 2> $OUT/bench_$i.err
done
python - <<PY
import json, statistics as st
rows=[json.loads(open("$OUT/bench_%d.json" % i).read().strip().splitlines()[-1]) for i in range(1, $N + 1)]
def col(f): return [f(d) for d in rows]
def line(name, v, fmt="%.2f"): print("| %s | %s | %s | %s | %.2f %% |" % (name, " / ".join(fmt % x for x in v), fmt % min(v), fmt % max(v), 100 * (max(v) - min(v)) / st.mean(v)))
print("| quantity | runs | min | max | spread |"); print("|---|---|---|---|---|")
line("armadillo_small steps/s", col(lambda d: d["value"]))
line("338 k tets steps/s", col(lambda d: d["at_scale"]["value"]))
line("2.7 M tets steps/s", col(lambda d: d["at_scale_large"]["value"]), "%.3f")
line("armadillo_small whole solve s (fresh process)", col(lambda d: d["end_to_end"]["cold"]["time_solve"]), "%.4f")
line("338 k tets whole solve s", col(lambda d: d["at_scale"]["end_to_end"]["time_solve"]), "%.3f")
line("2.7 M tets whole solve s", col(lambda d: d["at_scale_large"]["end_to_end"]["time_solve"]), "%.2f")
line("338 k tets constructor s", col(lambda d: d["at_scale"]["end_to_end"]["constructor_seconds"]), "%.3f")
PY
