#!/bin/bash
# kernel stats of scripts/solve_ab.py for library variants: bash scripts/gpu_solve_ab.sh "default fake" [config]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  if [ "$v" = "default" ]; then unset SANM_HIP_LIBRARY; else export SANM_HIP_LIBRARY=$ROOT/sanm_amd/libsanm_hip_$v.so; fi
  OUT=$ROOT/gpurun_out/solve_ab_$v
  rm -rf $OUT; mkdir -p $OUT
  (cd $ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 scripts/solve_ab.py ${2:-armadillo_small} 200 > $OUT/log.txt 2>&1)
  echo "== $v"; grep "^n " $OUT/log.txt
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
import re
agg={}
for r in rows:
    n=re.sub(r"[<(].*","",r["Name"].replace("void ","").replace("sanm_hip::(anonymous namespace)::","").replace("sanm_hip::",""))
    a=agg.setdefault(n,[0,0.0]); a[0]+=int(r["Calls"]); a[1]+=float(r["TotalDurationNs"])
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:6]:
    print("%-34s calls %6d  total %8.2f ms  avg %7.2f us"%(n,c,t/1e6,t/c/1e3))
PY
done
