#!/usr/bin/env python3
"""Latency of the calls the application really makes -- seconds-long clips and millisecond selections -- through the
host-pointer entry points (samples in pageable host memory in, results in host memory out), p50 / p95 over repeated
calls, each beside the C oracle's time for the same call on one host core.

  python tools/latency_bench.py [--out profiles/r03_latency.json] [--reps 40]

Call sites in the reference: the display path spectrogram.esm.js:1450 (compute_spectrogram_u8 on a clip), the
selection's power spectrum powerSpectrum.js:49 (compute_power_spectrum on a few thousand samples) and the detector's
analysis spectrogram of one region batCallDetector.js:1315 (compute_spectrogram at hop N/32 on some milliseconds).
The oracle (test infrastructure) is only the timed comparison here, never part of the product path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RATE = 384000


def _percentiles(samples_ms):
    a = np.sort(np.asarray(samples_ms))
    return {"p50_ms": round(float(np.percentile(a, 50)), 4), "p95_ms": round(float(np.percentile(a, 95)), 4),
            "min_ms": round(float(a[0]), 4), "calls": int(a.size)}


def _time_calls(fn, reps, warm=3, budget_s=20.0):
    for _ in range(warm):
        fn()
    out = []
    t_end = time.perf_counter() + budget_s
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e3)
        if time.perf_counter() > t_end and len(out) >= 5:
            break
    return out


def _oracle_ms(fn, budget_s=6.0, max_reps=5):
    fn()
    out = []
    t_end = time.perf_counter() + budget_s
    for _ in range(max_reps):
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e3)
        if time.perf_counter() > t_end:
            break
    return round(float(np.median(out)), 4)


def measure(reps=40, quick=False, with_oracle=True):
    """-> dict of cases.  quick: the short clips and the two selection-sized calls only (what bench.py adds to its line)."""
    import spectrogram_amd as sa
    from spectrogram_amd import detector, synth
    so = None
    if with_oracle:
        from oracle import detector_oracle, sono_oracle as so
    cases = {}
    e = sa.SpectrogramEngine(1024, "hann")
    oe = so.OracleEngine(1024, "hann") if so else None
    clips = [0.5, 2.0] if quick else [0.5, 2.0, 10.0, 60.0]
    for sec in clips:
        n = int(sec * RATE)
        x = synth.chirp_numpy(0, n, RATE, stream=3)
        r = _percentiles(_time_calls(lambda: e.compute_spectrogram_u8(x, 768, 20.0, 80.0), reps if sec <= 10 else max(8, reps // 4)))
        r["frames"] = int(sa.frames_for(n, 1024, 768))
        if oe is not None and sec <= 10.0:
            r["oracle_ms"] = _oracle_ms(lambda: oe.compute_spectrogram_u8(x, 768, 20.0, 80.0))
        cases[f"stft_u8_{sec:g}s_clip"] = r
    # a selection's power spectrum: 6 000 samples, FFT 1024, 75 % overlap (powerSpectrum.js:24-62)
    sel = synth.chirp_numpy(0, 6000, RATE, stream=5)
    r = _percentiles(_time_calls(lambda: sa.compute_power_spectrum(sel, RATE, 1024, "hann", 75.0), reps))
    if so:
        r["oracle_ms"] = _oracle_ms(lambda: so.compute_power_spectrum(sel, RATE, 1024, "hann", 75.0))
    cases["power_spectrum_6000_samples"] = r
    # the detector's analysis spectrogram of one 20 ms region at hop 32 (batCallDetector.js:1296-1377)
    roi = synth.chirp_numpy(0, int(0.020 * RATE), RATE, stream=7)
    r = _percentiles(_time_calls(lambda: detector.generate_spectrogram(e, roi, RATE, 10.0, 128.0), reps))
    r["frames"] = int(sa.frames_for(roi.size, 1024, 1024 - 32))
    if so:
        r["oracle_ms"] = _oracle_ms(lambda: detector_oracle.generate_spectrogram(oe, roi, RATE, 10.0, 128.0))
    cases["generate_spectrogram_20ms_roi"] = r
    # engine creation (window, tables, stream): paid once per engine, e.g. per changed FFT size in the application
    def create():
        q = sa.SpectrogramEngine(1024, "hann")
        q.free()
    cases["engine_create_destroy"] = _percentiles(_time_calls(create, min(reps, 20)))
    e.free()
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    res = {"sample_rate": RATE, "fft": 1024, "note": "host-pointer calls, pageable host memory in and out; oracle_ms = the C oracle "
           "(one host core) on the same call", "cases": measure(args.reps, args.quick)}
    text = json.dumps(res, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
