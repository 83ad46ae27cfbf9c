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
        # bench.py's ms_per_step is the mean of the steps after the warm-up: the trace pass runs 3 + 20
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows][3:]
        if t:
            print(f"the {len(t)} dispatches after the first three (bench.py's timed steps): avg {sum(t)/len(t)/1e6:.4f} ms")
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

# HBM traffic per dispatch for bench.py's roofline.traffic.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.
# On gfx950 FETCH_SIZE tallies each 128-byte read request as 64 bytes, so read bytes = 2 x FETCH_SIZE; that the
# factor applies to this kernel's loads is confirmed by the result itself: without it the kernel would have
# fetched less than the samples it provably reads once.  WRITE_SIZE is exact.
import hashlib
import json

# The counters describe ONE build of the library: its hash goes into the files, and bench.py only quotes them
# beside a timing of that same build (a rebuilt kernel would otherwise keep reporting the old counters).
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = os.environ.get("SONO_LIB_PATH") or os.path.join(_root, "spectrogram_amd", "lib", "libsonostft.so")
try:
    with open(_lib, "rb") as _fh:
        LIB_SHA256 = hashlib.sha256(_fh.read()).hexdigest()
except OSError:
    LIB_SHA256 = None

vals = {}
for f in find("*/**/*counter_collection.csv"):
    acc, disp = {}, {}
    for r in csv.DictReader(open(f)):
        if KEY in r.get("Kernel_Name", ""):
            c = r["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
            disp.setdefault(c, set()).add(r["Dispatch_Id"])
    for c in acc:
        vals[c] = acc[c] / max(len(disp[c]), 1)
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    read_b, write_b = 2.0 * vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    out_json = {"fft_size": 1024, "noverlap": 768, "seconds": 3600, "fetch_size_kib": vals["FETCH_SIZE"],
                "write_size_kib": vals["WRITE_SIZE"], "read_bytes": read_b, "write_bytes": write_b,
                "hbm_bytes": read_b + write_b, "lib_sha256": LIB_SHA256,
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per dispatch of the STFT kernel; "
                          "reads = 2 x FETCH_SIZE (gfx950 counts a 128-B request as 64 B), writes = WRITE_SIZE"}
    with open(os.path.join(out, "hbm_traffic.json"), "w") as fh:
        json.dump(out_json, fh, indent=1)
    print()
    print(f"== HBM traffic per dispatch: read {read_b/1e9:.3f} GB + write {write_b/1e9:.3f} GB = {(read_b+write_b)/1e9:.3f} GB ==")

# Vector-ALU counters per dispatch for bench.py's `valu_issue` object.  SQ_* cycle counters are in quad-cycles
# summed over the waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
if "SQ_INSTS_VALU" in vals and "SQ_WAVE_CYCLES" in vals:
    cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                       # shader cycles the dispatch took
    vc = {"insts_valu": vals["SQ_INSTS_VALU"], "active_inst_valu_quadcycles": vals.get("SQ_ACTIVE_INST_VALU"),
          "wave_quadcycles": vals["SQ_WAVE_CYCLES"], "wait_inst_any_quadcycles": vals.get("SQ_WAIT_INST_ANY"),
          "wait_any_quadcycles": vals.get("SQ_WAIT_ANY"), "active_inst_any_quadcycles": vals.get("SQ_ACTIVE_INST_ANY"),
          "waves": vals.get("SQ_WAVES"), "dispatch_cycles": cyc, "lib_sha256": LIB_SHA256,
          "source": "rocprofv3 --pmc passes of tools/profile.sh over bench.py, mean per dispatch of the STFT kernel"}
    if cyc > 0:
        # one vector instruction per 2 cycles and SIMD is the rate plain add / mul / fma instructions issue at
        vc["issue_busy_at_2_cycles"] = vals["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
    with open(os.path.join(out, "valu_counters.json"), "w") as fh:
        json.dump(vc, fh, indent=1)
    print(f"== vector ALU: {vals['SQ_INSTS_VALU']/1e9:.3f} G instructions per dispatch, "
          f"issue busy {vc.get('issue_busy_at_2_cycles', float('nan')):.3f} of one instruction per 2 cycles and SIMD ==")
