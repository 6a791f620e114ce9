set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $2 > $OUT/log.txt 2>&1
cd $ROOT
python - <<PY
import csv,glob,re,collections
rows=list(csv.DictReader(open(glob.glob("$OUT/stats/*kernel_stats.csv")[0])))
fam=collections.OrderedDict()
for r in rows:
    n=re.sub(r"^void ","",r["Name"]); n=n.replace("sanm_hip::(anonymous namespace)::","").replace("sanm_hip::",""); n=re.sub(r"[<(].*","",n)
    if n.startswith("spec_pass"): n="taylor"
    f=fam.setdefault(n,[0,0.0]); f[0]+=int(r["Calls"]); f[1]+=float(r["TotalDurationNs"])
tot=sum(v[1] for v in fam.values())
for k,(c,t) in sorted(fam.items(), key=lambda kv:-kv[1][1])[:14]: print(f"{k:36s} {c:6d} {t/1e6:9.2f} ms {t/c/1e3:9.1f} us {100*t/tot:5.1f}%")
PY
tail -c 300 $OUT/log.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
