#!/usr/bin/env python3
"""Compiles one HIP source of csrc for gfx950 and prints, per kernel, registers, spills, LDS and occupancy.

  python tools/resusage.py sono_tile_u8.hip [-DSONO_...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = sys.argv[2:]
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "spectrogram_amd", "csrc", src), "-o", "/dev/null"] + flags
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = d.replace("sono::(anonymous namespace)::", "").replace("void ", ""); d = re.sub(r"\(sono::StftArgs.*", "", d)
        cur = {"name": d}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print("%-70s vgpr %4s agpr %3s spill %3s sgpr %3s lds %7s occ %s" % (
        r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs"), r.get("LDS Size [bytes/block]"),
        r.get("Occupancy [waves/SIMD]")))
