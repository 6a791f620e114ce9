#!/bin/bash
# PMC counters of the solve level kernels (through gpurun): one rocprofv3 pass per counter group (kernel trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_levels}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/g$i.log 2>&1
  find $OUT/g$i -name "*.db" -delete
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, re
out="$OUT"
for g in sorted(glob.glob(out+"/g*/")):
    f=glob.glob(g+"/**/*counter_collection.csv", recursive=True)
    if not f: print(g, "no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"]
        if "level" not in n and "spec_pass4" not in n and "update_kernel" not in n: continue
        key=re.sub(r"^void ","",n).replace("sanm_hip::mfk::","")
        key=re.sub(r"\(.*","",key)+" grid="+r["Grid_Size"]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        if len(next(iter(agg[k].values())))<20: continue
        print(k, {c: round(sum(v)/len(v),1) for c,v in agg[k].items()}, "n=",len(next(iter(agg[k].values()))))
PY
