#!/bin/bash
# Recorded sweep of the dissection's leaf size (SANM_MF_LEAF) x the merged tree levels (SANM_MF_MERGE) on the three
# BASELINE-size workloads: leaf size trades launches per solve against fill.  One box, one session.
# usage (through gpurun): bash scripts/sweep_leaf_merge.sh <tag>   -> gpurun_out/<tag>/sweep.jsonl + sweep.md
set -u
TAG=${1:-sweep}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
: > $OUT/sweep.jsonl
for w in ${WORKLOADS:-armadillo_small bob human_arap16}; do
  for leaf in ${LEAVES:-16 32 64}; do
    for merge in ${MERGES:-none 1 1,3 1,3,5 1,2,3}; do
      SANM_MF_LEAF=$leaf SANM_MF_MERGE=$merge timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | \
        python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print(json.dumps({'workload':'$w','leaf':$leaf,'merge':'$merge','error':str(e)})); sys.exit(0)
f=d['roofline_families']; st=d['config']['solver_stats']
print(json.dumps({'workload':'$w','leaf':$leaf,'merge':'$merge','steps_per_s':d['value'],'ms_per_step':d['ms_per_step'],
  'solve_ms':f['solve']['ms_per_step'],'factor_ms':f['factor']['ms_per_step'],'solve_launches':f['solve']['launches_per_step'],
  'factor_launches':f['factor']['launches_per_step'],'solve_frac':f['solve']['frac'],'nr_level':st['nr_level'],'nr_front':st['nr_front'],
  'factor_nnz':st['factor_nnz'],'factor_gflop':st['factor_flops']/1e9,'max_front':st['max_front']}))" >> $OUT/sweep.jsonl
      tail -1 $OUT/sweep.jsonl
    done
  done
done
python - <<PY > $OUT/sweep.md
import json
rows=[json.loads(l) for l in open("$OUT/sweep.jsonl")]
print("| workload | leaf | merge | steps/s | ms/step | solve ms | factor ms | solve launches | factor launches | levels | fronts | factor nnz (M) | factor GFLOP |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    if 'error' in r: print(f"| {r['workload']} | {r['leaf']} | {r['merge']} | error: {r['error']} |"); continue
    print(f"| {r['workload']} | {r['leaf']} | {r['merge']} | {r['steps_per_s']:.1f} | {r['ms_per_step']:.3f} | {r['solve_ms']:.3f} | {r['factor_ms']:.3f} | {r['solve_launches']:.0f} | {r['factor_launches']:.0f} | {r['nr_level']} | {r['nr_front']} | {r['factor_nnz']/1e6:.2f} | {r['factor_gflop']:.2f} |")
PY
cat $OUT/sweep.md
