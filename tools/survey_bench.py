#!/usr/bin/env python3
"""Whole-file detection over many SHORT recordings on one GPU: survey.detect_units with 1, 2, 4, 8 units in flight.

  python tools/survey_bench.py [files] [seconds per file]        (default 200 files of 5 s at 384 kHz, 16-bit)
Prints one JSON line per setting: files per second, audio seconds per second, calls found.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_amd import survey, synth                                   # noqa: E402

files = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
sr = 384000
n = int(seconds * sr)
base = [np.clip(np.round(synth.chirp_numpy(0, n, sr, stream=s) * 32768.0), -32768, 32767).astype(np.int16) for s in range(8)]
units = [(i, 0, base[i % 8]) for i in range(files)]
survey.detect_units(units[:8], sr, config={"debugMode": True}, in_flight=8)          # first calls: engines' memory, code objects
want = None
for k in (1, 2, 4, 8, 1):
    t0 = time.perf_counter()
    table, rejected = survey.detect_units(units, sr, config={"debugMode": True}, in_flight=k)
    dt = time.perf_counter() - t0
    if want is None:
        want = table
    assert not rejected and np.array_equal(table.view(np.uint64), want.view(np.uint64))
    print(json.dumps({"in_flight": k, "files": files, "seconds_per_file": seconds, "wall_s": round(dt, 4), "files_per_s": round(files / dt, 1),
                      "audio_s_per_s": round(files * seconds / dt, 1), "ms_per_file": round(1e3 * dt / files, 3), "calls": int(table.shape[0])}), flush=True)
