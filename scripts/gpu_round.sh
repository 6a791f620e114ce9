#!/bin/bash
# One GPU session: GPU tests, the default bench line, optionally the profile set.
# usage (through gpurun): bash scripts/gpu_round.sh <tag> [tests|notests] [prof|noprof]
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "${2:-tests}" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 6000 $OUT/bench.json
if [ "${3:-noprof}" = "prof" ]; then
  bash scripts/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
fi
