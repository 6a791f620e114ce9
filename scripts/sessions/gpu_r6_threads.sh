#!/bin/bash
# end-to-end figure of a leg against the host thread cap      usage: gpu_r6_threads.sh <tag> <workload> <caps...>
set -u
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for TH in "$@"; do
  SANM_HOST_THREADS=$TH timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/bench_$TH.json 2> $OUT/bench_$TH.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$TH.json").read().strip().splitlines()[-1])
e=d["end_to_end"]
print("threads $TH", "e2e cold", round(e["cold"]["time_solve"],4), "cached", round(e["cached"]["time_solve"],4), e["iter"], {k:v for k,v in e["cached"]["setup_seconds"].items() if isinstance(v,float)})
PY
done
