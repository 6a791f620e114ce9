"""Turn gpurun_out/prof_<tag>/{stats,fetch,write} (scripts/collect_profiles.sh) into the committed summaries:
   profiles/<tag>_kernel_stats.{csv,md}, profiles/<tag>_pmc_traffic.md, profiles/pmc_traffic.json

   python scripts/summarize_profiles.py r01b [steps]      (default: the number of assemble_kernel launches = expansions)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: counted below from the trace (one assembly per expansion)
workload = sys.argv[3] if len(sys.argv) > 3 else "armadillo_small"
order = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    """kernel family: namespaces and template arguments stripped"""
    name = re.sub(r"^void ", "", name)
    if name.startswith("spec_pass"):  # the pass kernels compiled at run time for one graph (graph.cpp: spec_source)
        return "taylor_pass_kernel"
    name = name.replace("sanm_hip::(anonymous namespace)::", "").replace("sanm_hip::", "")
    name = re.sub(r"[<(].*", "", name)
    return name


# ---- kernel stats -----------------------------------------------------------
stats_csv = glob.glob(src + "/stats/*kernel_stats.csv")[0]
shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats_csv)))
fam = collections.OrderedDict()
for r in rows:
    f = fam.setdefault(short(r["Name"]), [0, 0.0])
    f[0] += int(r["Calls"])
    f[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in fam.values())
with open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w") as fo:
    if not steps:
        steps = next((c for k, (c, t) in fam.items() if k in ("assemble3_kernel", "assemble_kernel")), 14)
    fo.write(f"# rocprofv3 --kernel-trace --stats of `bench.py --workload {workload} --no-end-to-end --at-scale-workload none --at-scale-large-workload none` ({tag})\n\n")
    fo.write(f"{workload}, order {order}, 1 MI355X.  Template instantiations of one kernel are\n"
             f"summed.  {steps} ANM steps in the run (timed + warm-up + the bench's 2 family-measurement steps: one\n"
             f"`assemble3_kernel` / `assemble_kernel` launch each) -> per-step column = total / {steps}.  Full per-instantiation table: `{tag}_kernel_stats.csv`.\n\n")
    fo.write(f"Total kernel time {tot / 1e6:.1f} ms.\n\n")
    fo.write("| kernel | calls | total ms | avg us | ms/step | % |\n|---|---|---|---|---|---|\n")
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        fo.write(f"| {k} | {c} | {t / 1e6:.2f} | {t / c / 1e3:.2f} | {t / 1e6 / steps:.3f} | {100 * t / tot:.1f} |\n")
print(open(os.path.join(dst, f"{tag}_kernel_stats.md")).read())

# ---- PMC traffic ------------------------------------------------------------
def counter(path, name):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(glob.glob(path + "/*counter_collection.csv")[0])):
        if r["Counter_Name"] != name:
            continue
        a = acc.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch = counter(src + "/fetch", "FETCH_SIZE")
write = counter(src + "/write", "WRITE_SIZE")
out = {"workload": workload, "order": order,
       "note": f"rocprofv3 PMC, FETCH_SIZE doubled per MI355X_MICROARCH.md; see {tag}_pmc_traffic.md", "kernels": {}}
with open(os.path.join(dst, f"{tag}_pmc_traffic.md"), "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only)\n\n")
    fo.write("Command: `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 4 "
             f"--warmup 1 --workload {workload} --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none`\n({workload}, order {order}, 1 MI355X; scripts/collect_profiles.sh). "
             "Counter unit: KiB per dispatch. Correction per\nMI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of the "
             "bytes of a coalesced stream on gfx950 -> doubled; WRITE_SIZE is\nexact (calibrated on axpby_kernel, which "
             "writes n+1 doubles per launch: armadillo_small 38,047 = 297.2 KiB, refine:armadillo_small:1 235,378 = 1838.9 KiB).\n\n")
    fo.write("| kernel | dispatches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB | corrected HBM-side MB per launch "
             "(2*FETCH + WRITE) |\n|---|---|---|---|---|\n")
    for k, (c, v) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
        f_avg = v / c
        w_avg = write.get(k, [1, 0.0])[1] / max(write.get(k, [1, 0.0])[0], 1)
        traffic = (2 * f_avg + w_avg) * 1024
        out["kernels"][k] = {"fetch_kib": f_avg, "write_kib": w_avg, "traffic_bytes_per_launch": traffic}
        fo.write(f"| {k} | {c} | {f_avg:.1f} | {w_avg:.1f} | {traffic / 1e6:.2f} |\n")
# kernel families as bench.py reports them (roofline_families); traffic per launch = dispatch-weighted mean
FAMILY_OF = {"mfk::fwd_level_sub_kernel": "solve", "mfk::fwd_level_tr_kernel": "solve", "mfk::fwd_level_kernel": "solve", "mfk::bwd_level_kernel": "solve",
             "mfk::fwd_top_kernel": "solve", "mfk::bwd_top_kernel": "solve", "mfk::root_solve_kernel": "solve",
             "mfk::fwd_big_kernel": "solve", "mfk::bwd_big_kernel": "solve", "mfk::fwd_prep_kernel": "solve",
             "permute_out_dot_kernel": "solve", "mfk::permute_out_kernel": "solve", "mfk::permute_in_kernel": "solve",
             "taylor_pass_kernel": "taylor", "gather_rows3_kernel": "io", "gather_rows_kernel": "io",
             "assemble_kernel": "asm", "assemble3_kernel": "asm", "nonfinite_kernel": "asm"}
famacc = collections.OrderedDict()
for k, (c, v) in fetch.items():
    name = FAMILY_OF.get(k, "factor" if k.startswith("mfk::") else "tail")
    a = famacc.setdefault(name, [0, 0.0])
    a[0] += c
    a[1] += out["kernels"][k]["traffic_bytes_per_launch"] * c
out["families"] = {k: {"dispatches": c, "traffic_bytes_per_launch": v / c} for k, (c, v) in famacc.items()}
# profiles/pmc_traffic.json: the headline workload at the top level (as before), every other one under "workloads"
path = os.path.join(dst, "pmc_traffic.json")
try:
    cur = json.load(open(path))
except (OSError, ValueError):
    cur = {}
if workload == "armadillo_small":
    out["workloads"] = cur.get("workloads", {})
    cur = out
else:
    cur.setdefault("workloads", {})[workload] = out
json.dump(cur, open(path, "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_traffic.md")).read())
