"""Matrix-core busy share of the kernels of one rocprofv3 --kernel-trace --pmc run (scripts/sessions/gpu_r6_mfma_pmc.sh):
   python scripts/mfma_pmc_summary.py <dir with the run's csv files> <workload> > profiles/<name>.md"""
import collections
import csv
import glob
import re
import sys

src, wl = sys.argv[1], sys.argv[2]


def short(n):
    n = re.sub(r"^void ", "", n).replace("sanm_hip::mfk::", "").replace("sanm_hip::(anonymous namespace)::", "")
    return re.sub(r"[<(].*", "", n)


acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0])):
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
dur = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0])):
    dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print(f"# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- bench.py --workload {wl} "
      "--steps 1 --warmup 1 (one pass, no trace domains)\n")
print("GRBM_GUI_ACTIVE comes summed over the 8 XCDs (checked against the kernel trace of the same run: value / 8 / duration = the\n"
      "clock column); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs.  busy = MFMA busy cycles / (1024 x GUI_ACTIVE / 8): the\n"
      "share of a launch's cycles in which a SIMD's matrix pipe is busy, averaged over the SIMDs.\n")
print("| kernel | dispatches | total ms | clock GHz | matrix pipe busy |")
print("|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]:
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    m = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    if not cyc or not m:
        continue
    print(f"| {k} | {cnt[k]} | {dur[k] / 1e6:.1f} | {cyc / dur[k]:.2f} | {m / (1024 * cyc):.3f} |")
