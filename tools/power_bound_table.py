#!/usr/bin/env python3
"""From the per-case summaries of tools/profile_variants.sh: vector instructions per millisecond, clock and issue rate of
every STFT instantiation side by side.  The clock is the dispatch's shader cycles (GRBM_GUI_ACTIVE / 8 XCDs, counter
pass) over the dominant kernel's average time (trace pass): two runs of the same case on one box.

  python tools/power_bound_table.py gpurun_out/variants > profiles/r03_power_bound.txt
"""
import glob
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/variants"
rows = []
for f in sorted(glob.glob(os.path.join(d, "*_summary.txt"))):
    t = open(f).read()
    m = re.search(r"vector ALU: ([\d.]+) M instructions per dispatch over ([\d.]+) M shader cycles: issue busy ([\d.]+)", t)
    k = re.search(r"dominant kernel avg ([\d.]+) ms", t)
    w = re.search(r"'Workgroup_Size_X': '(\d+)'", t)
    v = re.search(r"'VGPR_Count': '(\d+)'", t)
    if not (m and k):
        continue
    insts, cycles, busy, ms = float(m.group(1)), float(m.group(2)), float(m.group(3)), float(k.group(1))
    name = os.path.basename(f)[: -len("_summary.txt")]
    rows.append((name, int(w.group(1)) // 64 if w else 0, ms, insts, cycles / ms, busy, cycles / ms * busy, insts / ms / 1e3))
print("Vector instructions retired per millisecond by the library's kernels, one box, tools/profile_variants.sh")
print("(clock = shader cycles of the counter pass / kernel time of the trace pass; busy = instructions x 2 cycles / SIMD-cycles)")
print()
print(f"{'case':24s} {'waves/WG':>8s} {'kernel ms':>10s} {'VALU M':>9s} {'clock GHz':>10s} {'busy':>6s} {'clock x busy':>13s} {'G instr/ms':>11s}")
for r in rows:
    print(f"{r[0]:24s} {r[1]:8d} {r[2]:10.3f} {r[3]:9.1f} {r[4]:10.2f} {r[5]:6.3f} {r[6]:13.2f} {r[7]:11.3f}")
