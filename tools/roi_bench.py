#!/usr/bin/env python3
"""Times the detector's whole-file pass (fast scan -> regions -> analysis -> framed calls, then the same with the
measurement stage) region by region (framing.process_full_file) and batched (framing.process_full_file_batched) on a
synthetic recording and checks that both give the same calls.  usage: python tools/roi_bench.py [seconds] [--skip-single]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectrogram_amd as sa                       # noqa: E402
from spectrogram_amd import framing as fr, synth   # noqa: E402


def main():
    seconds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 120
    sr = 384000
    x = synth.chirp_torch(0, seconds * sr, sr, stream=0, device="cuda").cpu().numpy()
    e = sa.SpectrogramEngine(1024, "hann")
    args = (e, x, sr, 15.0, 110.0, {"threshold_dB": -50.0}, {})
    fr.process_full_file_batched(e, x[: 2 * sr], sr, 15.0, 110.0, {"threshold_dB": -50.0}, {})      # warm-up
    t0 = time.perf_counter()
    batched = fr.process_full_file_batched(*args)
    t1 = time.perf_counter()
    rois = len({c["roi"] for c in batched})
    print(f"batched: {len(batched)} calls in {rois} regions, {t1 - t0:.2f} s for {seconds} s of audio "
          f"({seconds / (t1 - t0):.0f} x real time)", flush=True)
    if "--skip-single" not in sys.argv:
        t0 = time.perf_counter()
        single = fr.process_full_file(*args)
        t1 = time.perf_counter()
        print(f"region by region: {len(single)} calls, {t1 - t0:.2f} s ({seconds / (t1 - t0):.0f} x real time)", flush=True)
        keys = ("roi", "startTime_s", "endTime_s", "duration_ms", "peakPower_dB", "peakTime_s", "highpass_kHz")
        same = len(single) == len(batched) and all(all(a[k] == b[k] for k in keys) for a, b in zip(single, batched))
        print("identical:", same, flush=True)
        if not same:
            sys.exit(1)
    # the measurement stage on top (debug mode on, so that calls on which the reference raises outside it are kept)
    margs = args[:-1] + ({"debugMode": True},)
    t0 = time.perf_counter()
    measured = fr.process_full_file_batched(*margs, measure=True)
    t1 = time.perf_counter()
    print(f"batched + measurement: {len(measured)} calls, {t1 - t0:.2f} s ({seconds / (t1 - t0):.0f} x real time)", flush=True)
    t0 = time.perf_counter()
    native = fr.process_full_file_native(*margs)
    t1 = time.perf_counter()
    print(f"one library call: {len(native)} calls, {t1 - t0:.2f} s ({seconds / (t1 - t0):.0f} x real time)", flush=True)
    strip = lambda c: {k: v for k, v in c.items() if k != "zonalNoiseMap"}
    if not (len(native) == len(measured) and all(a == strip(b) for a, b in zip(native, measured))):
        print("one library call: DIFFERENT", flush=True)
        sys.exit(1)
    if "--skip-single" not in sys.argv:
        t0 = time.perf_counter()
        single = fr.process_full_file(*margs, measure=True)
        t1 = time.perf_counter()
        print(f"region by region + measurement: {len(single)} calls, {t1 - t0:.2f} s ({seconds / (t1 - t0):.0f} x real time)", flush=True)
        strip = lambda c: {k: v for k, v in c.items() if k != "zonalNoiseMap"}
        same = len(single) == len(measured) and all(strip(a) == strip(b) for a, b in zip(single, measured))
        print("identical:", same, flush=True)
        if not same:
            sys.exit(1)


if __name__ == "__main__":
    main()
