#!/bin/bash
# kernel trace of scripts/ctor_sync_probe.py: the long kernels of the constructors     usage: gpu_r5_probe_trace.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/scripts/ctor_sync_probe.py > $OUT/probe.log 2>&1
cd $ROOT
grep solver $OUT/probe.log
python - <<PY
import csv, glob
rows=list(csv.DictReader(open(glob.glob("$OUT/trace/*kernel_trace.csv")[0])))
t0=min(int(r["Start_Timestamp"]) for r in rows)
big=[(int(r["Start_Timestamp"])-t0, int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Kernel_Name"][:60]) for r in rows if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>500000]
for s,d,n in sorted(big): print(f"{s/1e6:10.3f} ms  {d/1e6:8.3f} ms  {n}")
PY
