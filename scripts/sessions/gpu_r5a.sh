#!/bin/bash
# round 5, session a: GPU suite, the default bench line (end_to_end + at_scale + CPU baseline), kernel stats of the
# at-scale workload.   usage (through gpurun): bash scripts/gpu_r5a.sh <tag> [tests|notests]
set -u
TAG=${1:-r5a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "${2:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  grep -E "passed|failed|rc=|Error|error" $OUT/pytest.log | tail -5
fi
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
cut -c1-400 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x8 -o run -- python3 $ROOT/bench.py --workload refine:armadillo_small:1 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/stats_x8.log 2>&1
ls $OUT/stats_x8/ | head
cd $ROOT
python scripts/prof_summary.py $OUT/stats_x8 > $OUT/kernel_stats_x8.md 2>/dev/null || true
head -30 $OUT/kernel_stats_x8.md
