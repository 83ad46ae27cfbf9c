#!/usr/bin/env python3
"""One case of tools/profile_variants.sh -> a text summary: vbench's own timing and algorithmic bytes, the
library kernels of the trace pass (per-kernel time statistics), and for the kernel that takes the most time
its counters per dispatch, the roofline fraction its duration gives and how busy its vector issue was."""
import csv
import glob
import json
import os
import sys

d, case = sys.argv[1], sys.argv[2]
SKIP = ("fill_noise", "to_pcm16")
# the kernel a case is about, where the case's set-up launches heavier ones
TARGET = {"power_db": "power_db_kernel", "db_max": "db_max_kernel", "db_active": "db_active_kernel",
          "frame_features": "frame_features_kernel", "zonal_floors": "zone_hist_kernel", "band_active_2pass": "band_active_kernel",
          "runs": "run_edges_kernel", "runs_static": "runs_emit_kernel", "pcm_to_f32": "pcm_to_f32_kernel",
          "filterbank_1024": "filterbank_taps_u8_kernel"}
# kernels of the case's set-up (launched once, before the timed calls), left out of the per-call kernel time
SETUP = {"power_db": ("stft_tile_kernel", "db_active_kernel"), "db_max": ("stft_tile_kernel", "power_db_kernel", "db_active_kernel"),
         "db_active": ("stft_tile_kernel", "power_db_kernel"), "frame_features": ("stft_tile_kernel", "power_db_kernel", "db_active_kernel"),
         "zonal_floors": ("stft_tile_kernel", "power_db_kernel", "db_active_kernel"),
         "band_active_2pass": ("stft_tile_kernel", "power_db_kernel", "db_active_kernel"),
         "runs": ("stft_tile_kernel", "power_db_kernel", "db_active_kernel"),
         "runs_static": ("stft_tile_kernel", "power_db_kernel", "db_active_kernel")}


def find(pattern):
    return sorted(glob.glob(os.path.join(d, pattern), recursive=True))


vb = {}
try:
    vb = json.load(open(os.path.join(d, "vbench.json")))
except Exception as err:                                     # noqa: BLE001
    print("no vbench line:", err)
print(f"case {case}: {vb.get('what', '?')}")
print(f"vbench (HIP events on the launch stream, {vb.get('reps')} launches): avg {vb.get('ms_avg')} ms, best {vb.get('ms_best')} ms per call")
print(f"algorithmic HBM bytes per call: {vb.get('algorithmic_bytes', 0) / 1e9:.4f} GB -> {vb.get('GBps_avg')} GB/s = "
      f"{vb.get('frac_of_8TBps')} of 8 TB/s at the average; {vb.get('units_per_s')} units/s")
print()
print("== kernel trace: the library's kernels (ns) ==")
top, total = None, 0.0
for f in find("trace/**/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if not any(s in r["Name"] for s in SKIP + SETUP.get(case, ()))]
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:8]:
        share = float(r["TotalDurationNs"]) / total if total else 0.0
        print(f"{r['Name'][:150]}\n    calls {r['Calls']}  avg {float(r['AverageNs']) / 1e3:.1f} us  min {float(r['MinNs']) / 1e3:.1f}  "
              f"max {float(r['MaxNs']) / 1e3:.1f}  share of the case's kernel time {share:.3f}")
    if rows:
        top = next((r for r in rows if TARGET.get(case, "\0") in r["Name"]), rows[0])
if top is None:
    sys.exit(0)
name = top["Name"]
for f in find("trace/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r.get("Kernel_Name", "") == name]
    if rows:
        r0 = rows[0]
        print()
        print("dominant kernel launch:", {k: r0[k] for k in ("Workgroup_Size_X", "Grid_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                            "LDS_Block_Size", "Scratch_Size") if k in r0})
vals = {}
for f in find("*/**/*counter_collection.csv"):
    acc, disp = {}, {}
    for r in csv.DictReader(open(f)):
        if r.get("Kernel_Name", "") == name:
            c = r["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
            disp.setdefault(c, set()).add(r["Dispatch_Id"])
    for c in acc:
        vals[c] = acc[c] / max(len(disp[c]), 1)
print()
print("== counters of the dominant kernel (mean per dispatch) ==")
for c in sorted(vals):
    print(f"{c:24s} {vals[c]:18.1f}")
avg_ms = float(top["AverageNs"]) / 1e6
calls_per_launch = max(1, round(int(top["Calls"]) / 13))       # 3 warm-up + 10 timed calls in the trace pass
print()
if vb.get("algorithmic_bytes"):
    frac = float(top["TotalDurationNs"]) / total if total else 1.0
    print(f"== roofline (HBM, 8 TB/s): dominant kernel avg {avg_ms:.4f} ms x {calls_per_launch} per call; the case's algorithmic bytes over "
          f"the kernel time of one call ({avg_ms * calls_per_launch / frac:.4f} ms): "
          f"{vb['algorithmic_bytes'] / (avg_ms * calls_per_launch / frac) / 1e6:.0f} GB/s = "
          f"{vb['algorithmic_bytes'] / (avg_ms * calls_per_launch / frac) / 1e6 / 8000:.3f} ==")
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    rd, wr = 2.0 * vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    print(f"== HBM traffic of the dominant kernel per dispatch: read {rd / 1e9:.3f} GB (2 x FETCH_SIZE) + write {wr / 1e9:.3f} GB = "
          f"{(rd + wr) / 1e9:.3f} GB; algorithmic {vb.get('algorithmic_bytes', 0) / 1e9:.3f} GB per call ==")
if "SQ_INSTS_VALU" in vals and vals.get("GRBM_GUI_ACTIVE", 0) > 0:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
    busy = vals["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
    wc = vals.get("SQ_WAVE_CYCLES", 0.0)
    parts = ", ".join(f"{k[3:].lower()} {vals[k] / wc:.3f}" for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in vals and wc)
    print(f"== vector ALU: {vals['SQ_INSTS_VALU'] / 1e6:.1f} M instructions per dispatch over {cyc / 1e6:.3f} M shader cycles: issue busy "
          f"{busy:.3f} of one instruction per 2 cycles and SIMD; wave time split: {parts} ==")
