#!/usr/bin/env python3
"""Capture golden vectors of the detector front end from the reference's own
JavaScript (fastScanSegmentsWasm, mergeAndPadSegments, generateSpectrogramWasm,
detectCallSegments of modules/batCallDetector.js) driving the reference wasm.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where /root/reference
and Node are present; the GPU box only ever sees the committed output,
tests/golden/golden_detector.npz (inputs + the reference's outputs, no code).

  python oracle/make_golden_detector.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from make_golden import chirp, tone, white  # noqa: E402

WASM = os.environ.get("SONO_REF_WASM", "/root/reference/modules/spectrogram_wasm_bg.wasm")
DETECTOR_JS = os.environ.get("SONO_REF_DETECTOR", "/root/reference/modules/batCallDetector.js")
OUT = os.path.join(ROOT, "tests", "golden", "golden_detector.npz")


def build_cases():
    cases, bufs, pos = [], [], [0]

    def add(name, x, sr, fft=1024, window="hann", flow=10.0, fhigh=128.0, scan_thr=-60.0, pad_ms=10.0,
            hop_percent=3.125, call_thr=-24.0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        cases.append(dict(name=name, fft=fft, window=window, sr=sr, flow_khz=flow, fhigh_khz=fhigh,
                          scan_threshold_db=scan_thr, padding_ms=pad_ms, hop_percent=hop_percent,
                          call_threshold_db=call_thr, x_off=pos[0], x_len=int(x.size)))
        bufs.append(x)
        pos[0] += x.size

    sr = 384000
    # bat-pass-like pulses in noise; the reference's analysis hop (3.125 %) on a short stretch
    add("pulses384k", chirp(14000, sr, 51, period=0.02, t0=0.004), sr, flow=20.0, fhigh=100.0)
    add("pulses384k_band30_60", chirp(46080, sr, 52, period=0.04, t0=0.01), sr, flow=30.0, fhigh=60.0, scan_thr=-50.0,
        hop_percent=12.5)
    # band reaching Nyquist: maxBin == bins, the scan reads into the next frame's first bin
    add("pulses_fullband", chirp(30000, sr, 53, period=0.03, t0=0.002), sr, flow=0.0, fhigh=192.0, scan_thr=-55.0,
        hop_percent=25.0)
    # call still running at the end of the buffer; coarser analysis hop
    tail = np.concatenate([1e-4 * white(9000, 54, 1.0), tone(5000, sr, 45000.0, amp=0.5)])
    add("open_tail", tail, sr, hop_percent=12.5, scan_thr=-40.0, flow=30.0, fhigh=60.0)
    # nothing above threshold
    add("quiet", 1e-5 * white(8000, 55, 1.0), sr, scan_thr=-30.0, hop_percent=25.0)
    # two bursts closer than the padding: merged into one region
    two = 1e-4 * white(40000, 56, 1.0)
    two[6000:8000] += tone(2000, sr, 50000.0, amp=0.4)
    two[11000:13000] += tone(2000, sr, 38000.0, amp=0.3)
    two[30000:31500] += tone(1500, sr, 60000.0, amp=0.2)
    add("bursts_merge", two, sr, pad_ms=10.0, call_thr=-18.0, hop_percent=12.5, flow=30.0, fhigh=70.0)
    # other sizes / rates
    add("pulses500k_512", chirp(30000, 500000, 57, period=0.02, t0=0.003), 500000, fft=512, flow=20.0, fhigh=100.0,
        hop_percent=12.5)
    add("pulses256k_256_hamming", chirp(20000, 256000, 58, period=0.02, t0=0.003), 256000, fft=256, window="hamming",
        flow=15.0, fhigh=120.0, hop_percent=12.5, scan_thr=-50.0)
    add("shorter_than_fft", tone(700, sr, 40000.0, amp=0.3), sr)
    return cases, np.concatenate(bufs)


def main():
    cases, inputs = build_cases()
    with tempfile.TemporaryDirectory() as work:
        json.dump(cases, open(os.path.join(work, "cases.json"), "w"))
        inputs.tofile(os.path.join(work, "inputs.bin"))
        node = ["node"]
        ver = subprocess.check_output(["node", "--version"]).decode().strip()
        if int(ver.lstrip("v").split(".")[0]) < 16:
            node += ["--experimental-wasm-mv", "--experimental-wasm-anyref"]
        subprocess.check_call(node + [os.path.join(HERE, "detector_driver.js"), WASM, DETECTOR_JS, work])
        man = json.load(open(os.path.join(work, "outputs.json")))
        blob = open(os.path.join(work, "outputs.bin"), "rb").read()
    dt = {"f32": np.float32, "f64": np.float64, "u8": np.uint8}
    arrays = {}
    for m in man:
        arrays[f"{m['case']}/{m['key']}"] = np.frombuffer(blob, np.dtype(dt[m["dtype"]]), m["count"], m["offset"]).copy()
    meta = []
    for c in cases:
        c = dict(c)
        arrays[f"{c['name']}/x"] = inputs[c["x_off"]:c["x_off"] + c["x_len"]].copy()
        c.pop("x_off"); c.pop("x_len")
        meta.append(c)
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **arrays)
    print(f"wrote {OUT}: {len(cases)} cases, {os.path.getsize(OUT)} bytes")
    for c in meta:
        n = c["name"]
        print(f"  {n}: scan {arrays[n + '/scan'].reshape(-1, 2).tolist()} merged {arrays[n + '/merged'].reshape(-1, 2).tolist()}"
              f" shape {arrays[n + '/shape'].tolist()} calls {arrays.get(n + '/calls', np.zeros(0)).reshape(-1, 2).tolist()}")


if __name__ == "__main__":
    sys.exit(main())
