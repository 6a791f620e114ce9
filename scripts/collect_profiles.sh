#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the two PMC passes of one bench leg.
# usage: bash scripts/collect_profiles.sh <tag> [workload] [stat steps] [pmc steps]   -> gpurun_out/prof_<tag>/{stats,fetch,write}
set -u
TAG=${1:-run}
WL=${2:-armadillo_small}
S1=${3:-12}
S2=${4:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
FLAGS="--workload $WL --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps $S1 --warmup 2 $FLAGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ROOT/bench.py --steps $S2 --warmup 1 $FLAGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ROOT/bench.py --steps $S2 --warmup 1 $FLAGS > $OUT/write.log 2>&1
# the kernel trace is large; keep the stats and the counter files
find $OUT -name "*.db" -delete
ls -la $OUT/*
