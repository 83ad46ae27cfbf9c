#!/usr/bin/env python3
"""Copies what one profiling call on the GPU box left under gpurun_out/ into profiles/ (the round's cfg2 counters and
trace summary, the workloads' bench lines, the whole-file detection's kernel table and counters).

  gpurun --timeout 1200 -- 'bash tools/profile_all.sh'      (on the GPU box)
  python tools/keep_profiles.py r03
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
shutil.copy("gpurun_out/keep/hbm_traffic.json", "profiles/hbm_traffic.json")
shutil.copy("gpurun_out/keep/valu_counters.json", "profiles/valu_counters.json")
shutil.copy("gpurun_out/keep/hbm_traffic.json", "profiles/%s_cfg2_hbm_traffic.json" % tag)
shutil.copy("gpurun_out/keep/valu_counters.json", "profiles/%s_cfg2_valu_counters.json" % tag)
shutil.copy("gpurun_out/prof/trace/trace_kernel_stats.csv", "profiles/%s_cfg2_kernel_stats.csv" % tag)
shutil.copy("gpurun_out/prof/summary.txt", "profiles/%s_cfg2_summary.txt" % tag)
shutil.copy("gpurun_out/w/cfg2.json", "profiles/%s_bench_default.json" % tag)
with open("profiles/%s_workloads.jsonl" % tag, "w") as out:
    for f in ("cfg2", "cfg2_pcm16", "cfg3", "cfg4", "cfg5_pcm16", "cfg5_1h", "cfg5_12h"):
        out.write(open("gpurun_out/w/%s.json" % f).read().strip().splitlines()[-1] + "\n")
shutil.copy("gpurun_out/prof19/summary.txt", "profiles/%s_cfg5_detection_counters.txt" % tag)
shutil.copy("gpurun_out/prof18/p_kernel_stats.csv", "profiles/%s_cfg5_detection_kernel_stats.csv" % tag)
rows = []
for r in csv.DictReader(open("gpurun_out/prof18/p_kernel_stats.csv")):
    k = r["Name"]
    if "sono" not in k:
        continue
    k = k.replace("sono::(anonymous namespace)::", "").split("(")[0]
    rows.append((k, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
rows.sort(key=lambda r: -r[2])
with open("profiles/%s_cfg5_detection_kernels.txt" % tag, "w") as out:
    out.write("kernels of the library in three whole-file detections of one hour of 16-bit audio each (bench.py --workload cfg5 "
              "--seconds 3600 --steps 2 --warmup 1 under rocprofv3 --kernel-trace --stats; SONO_NO_MEASURE_OVERLAP=1, i.e. one "
              "stage at a time: with a group's measurement beside the next group's analysis the kernels share the card and "
              "their durations no longer add up)\n")
    out.write("%-56s %6s %12s %10s %14s\n" % ("kernel", "calls", "total ms", "avg ms", "ms / detection"))
    for k, c, t, a in rows:
        out.write("%-56s %6d %12.2f %10.3f %14.2f\n" % (k[:56], c, t, a, t / 3))
    out.write("sum %.2f ms / detection\n" % (sum(r[2] for r in rows) / 3))
print(open("profiles/%s_cfg5_detection_kernels.txt" % tag).read())
for line in open("profiles/%s_workloads.jsonl" % tag):
    d = json.loads(line)
    print(d["config"].get("name"), round(d["ms_per_step"], 2), d.get("host_rss_peak_gb"), d.get("ms_per_hour_of_audio"), d.get("lib_sha256", "")[:12],
          d["roofline"].get("traffic") if "roofline" in d else None)
