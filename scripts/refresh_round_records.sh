#!/bin/bash
# profiles/r<NN>_* from one gpu_r6_final.sh session (gpurun_out/<tag>, gpurun_out/prof_<tag>{,_x8,_x64})    usage: refresh_round_records.sh <tag> <NN>
set -eu
T=$1; R=r$2
cd "$(dirname "$0")/.."
python scripts/summarize_profiles.py $T > /dev/null
python scripts/summarize_profiles.py ${T}_x8 0 refine:armadillo_small:1 20 > /dev/null
cd profiles
for f in ${T}_kernel_stats.csv ${T}_kernel_stats.md ${T}_pmc_traffic.md; do n=${f/$T/$R}; sed "s/${T}_/${R}_/g; s/(${T})/(${R}, scripts\/sessions\/gpu_r6_final.sh)/" $f > $n; rm $f; done
for f in ${T}_x8_kernel_stats.csv ${T}_x8_kernel_stats.md ${T}_x8_pmc_traffic.md; do n=${f/$T/$R}; sed "s/${T}_x8_/${R}_x8_/g; s/(${T}_x8)/(${R}, scripts\/sessions\/gpu_r6_final.sh)/" $f > $n; rm $f; done
sed -i "s/${T}_x8_/${R}_x8_/g; s/${T}_/${R}_/g" pmc_traffic.json
cd ..
python - "$T" "$R" <<'PY'
import csv, glob, collections, re, sys
T, R = sys.argv[1], sys.argv[2]
src=f"gpurun_out/prof_{T}_x64/stats"
def short(name):
    name = re.sub(r"^void ", "", name)
    if name.startswith("spec_pass"): return "taylor_pass_kernel"
    name = name.replace("sanm_hip::(anonymous namespace)::", "").replace("sanm_hip::", "")
    return re.sub(r"[<(].*", "", name)
rows=list(csv.DictReader(open(glob.glob(src+"/*kernel_stats.csv")[0])))
fam=collections.OrderedDict()
for r in rows:
    f=fam.setdefault(short(r["Name"]),[0,0.0]); f[0]+=int(r["Calls"]); f[1]+=float(r["TotalDurationNs"])
tot=sum(v[1] for v in fam.values())
steps=next(c for k,(c,t) in fam.items() if k in ("assemble3_kernel","assemble_kernel"))
with open(f"profiles/{R}_kernel_stats_x64.md","w") as fo:
    fo.write(f"# rocprofv3 --kernel-trace --stats of `bench.py --workload refine:armadillo_small:2 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none` ({R}, scripts/sessions/gpu_r6_final.sh)\n\n")
    fo.write(f"refine:armadillo_small:2 (2.7 M tets, 1.62 M unknowns), order 20, 1 MI355X.  Template instantiations of one kernel are summed.  {steps} ANM steps in the run -> per-step column = total / {steps}.\n\n")
    fo.write(f"Total kernel time {tot/1e6:.1f} ms.\n\n| kernel | calls | total ms | avg us | ms/step | % |\n|---|---|---|---|---|---|\n")
    for k,(c,t) in sorted(fam.items(), key=lambda kv:-kv[1][1]):
        fo.write(f"| {k} | {c} | {t/1e6:.2f} | {t/c/1e3:.2f} | {t/1e6/steps:.3f} | {100*t/tot:.1f} |\n")
PY
for f in bench_bob bench_human_arap16 bench_block_32 bench_block_48; do tail -1 gpurun_out/$T/$f.json > profiles/${R}_$f.json; done
tail -1 gpurun_out/$T/bench.json > profiles/${R}_bench.json
for f in gpurun_out/$T/parity_steps_*.json; do cp $f profiles/${R}_$(basename $f); done
git status --short profiles | head -20
