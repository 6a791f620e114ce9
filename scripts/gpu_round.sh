#!/bin/bash
# One GPU session: GPU tests, the default bench line (with the CPU baseline), optionally the profile set and the
# bench lines of the other BASELINE-size workloads.
# usage (through gpurun): bash scripts/gpu_round.sh <tag> [tests|notests] [prof|noprof] [all|one]
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "${2:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  grep -E "passed|failed|rc=" $OUT/pytest.log | tail -3
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cut -c1-300 $OUT/bench.json
if [ "${4:-one}" = "all" ]; then
  for w in bob human_arap16 block:32 block:48; do
    timeout 900 python bench.py --steps 10 --warmup 3 --workload $w --no-cpu-baseline > $OUT/bench_${w/:/}.json 2>> $OUT/bench.err
    cut -c1-200 $OUT/bench_${w/:/}.json
  done
fi
if [ "${3:-noprof}" = "prof" ]; then
  bash scripts/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
  ls $ROOT/gpurun_out/prof_$TAG/
fi
