"""Summarise a rocprofv3 --kernel-trace --stats --output-format csv directory:
   python scripts/prof_summary.py gpurun_out/profNN [steps]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms" + (f" = {tot / 1e6 / steps:.2f} ms/step" if steps else ""))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs']) / 1e6:8.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:8.2f} pct={float(r['Percentage']):.1f}")
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ["fwd_level", "bwd_level"]:
    by = collections.OrderedDict()
    for r in rows:
        if name in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("::")[-1][:24], int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]),
                   "lds", r["LDS_Block_Size"], "vgpr", r["VGPR_Count"], "scr", r["Scratch_Size"])
            by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tt = 0
    for k, v in by.items():
        print("   ", k, "n", len(v), "avg_us %.2f" % (sum(v) / len(v)))
        tt += sum(v) / len(v)
    print("   sum of per-level averages: %.1f us" % tt)
