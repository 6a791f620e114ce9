#!/bin/bash
# kernel-trace timelines of the default build and of the variants named by env settings (through gpurun):
#   bash scripts/trace_ab.sh <tag> "VAR=1" "VAR2=1" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for v in "" "$@"; do
  OUT=$ROOT/gpurun_out/${TAG}_v$i
  mkdir -p $OUT
  if [ -n "$v" ]; then export "$v"; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
  if [ -n "$v" ]; then unset "${v%%=*}"; fi
  python $ROOT/scripts/step_timeline.py $OUT/trace 3 --all > $OUT/timeline_all.txt 2>&1
  echo "== variant $i: $v"; grep -E "spec_pass|step of" $OUT/timeline_all.txt | head -40
  find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
  i=$((i+1))
done
