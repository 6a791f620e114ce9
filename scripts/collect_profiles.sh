#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the two PMC passes of the bench command.
# usage: bash scripts/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write}
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
# the kernel trace is large; keep the stats and the counter files
find $OUT -name "*.db" -delete
ls -la $OUT/*
