#!/usr/bin/env python3
"""Case definitions (seeded inputs) of tests/golden/golden_detector.npz: the detector
front end (fastScanSegmentsWasm, mergeAndPadSegments, generateSpectrogramWasm,
detectCallSegments; batCallDetector.js:878-949, 1028-1056, 1296-1377, 1540-1598).

TEST INFRASTRUCTURE ONLY.  The expected outputs in the fixture were recorded once
from the reference's own methods; the fixture is frozen data and is never
regenerated -- nothing here loads or executes anything that ships inside the
reference.  build_cases() is the record of the inputs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from make_golden import chirp, tone, white  # noqa: E402

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
    """The fixture is frozen data: nothing here executes anything that ships inside the
    reference.  build_cases() remains as the record of the inputs the fixture was made from."""
    cases = build_cases()
    n = len(cases[0]) if isinstance(cases, tuple) else len(cases)
    print(f"{n} case definitions; tests/golden/golden_detector.npz is frozen and is not regenerated "
          "(its expected outputs were recorded once from the reference's own code paths; see the module docstring)")
    return 1


if __name__ == "__main__":
    sys.exit(main())
