#!/bin/bash
# HIP API trace of one cold whole solve      usage: gpu_r6_hiptrace.sh <tag> [workload]
set -u
TAG=$1; WL=${2:-armadillo_small}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/scripts/cold_solve.py $WL > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-600
find $OUT/trace -name "*.db" -delete
ls $OUT/trace/* | head
python3 - <<PY
import csv, glob
f=glob.glob("$OUT/trace/**/*hip_api_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:25]: print(f'{r["Name"]:45s} calls={r["Calls"]:>7s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.1f} max_us={float(r["MaxNs"])/1e3:9.1f}')
PY
