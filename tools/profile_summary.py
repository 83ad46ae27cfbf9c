#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile.sh into a short text summary:
per-kernel time statistics of the trace pass and the counter values of the STFT
kernel (averaged over its dispatches)."""
import csv
import glob
import os
import sys

out = sys.argv[1]
KEY = "stft_"


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel trace: per-kernel statistics (ns) ==")
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")})
print()
print("== kernel trace: the STFT kernel's dispatches ==")
for f in find("trace/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if KEY in r.get("Kernel_Name", "")]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if d:
        r0 = rows[0]
        print(f"dispatches {len(d)}  avg {sum(d)/len(d)/1e6:.4f} ms  min {min(d)/1e6:.4f} ms  max {max(d)/1e6:.4f} ms")
        print({k: r0[k] for k in r0 if k in ("Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "VGPR_Count", "Accum_VGPR_Count",
                                            "SGPR_Count", "LDS_Block_Size", "Scratch_Size")})
print()
print("== counters of the STFT kernel (mean per dispatch) ==")
for f in find("*/**/*counter_collection.csv"):
    acc = {}
    n = {}
    for r in csv.DictReader(open(f)):
        if KEY not in r.get("Kernel_Name", ""):
            continue
        c = r["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
        n[c] = n.get(c, 0) + 1
    disp = {}
    for r in csv.DictReader(open(f)):
        if KEY in r.get("Kernel_Name", ""):
            disp.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
    for c in sorted(acc):
        k = max(len(disp[c]), 1)
        print(f"{os.path.basename(os.path.dirname(os.path.dirname(f))) if False else ''}{c:28s} {acc[c]/k:18.1f}   (dispatches {k})")
