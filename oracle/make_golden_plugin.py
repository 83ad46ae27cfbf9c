#!/usr/bin/env python3
"""Case definitions (seeded inputs) of tests/golden/golden_plugin.json: the spectrogram
plugin's frame contract (getFrequencies, spectrogram.esm.js:1364-1529).  TEST
INFRASTRUCTURE ONLY.  The expected outputs were recorded once from the reference's own
method; the fixture is frozen data and is never regenerated -- nothing here loads or
executes anything that ships inside the reference.  build_cases() is the record of the
inputs.
"""
import base64
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from make_golden import chirp, tone, white  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_plugin.json")


def build_cases():
    cases, bufs, pos = [], [], [0]

    def add(name, channels, sr, fft, noverlap=0, canvas_width=800, peak_mode=True, peak_threshold=0.4,
            split=False, gain=20.0, rng=80.0, window="hann", scale="linear", fmin=0, fmax=0):
        chs = []
        for x in channels:
            x = np.ascontiguousarray(x, np.float32)
            chs.append([pos[0], int(x.size)])
            bufs.append(x)
            pos[0] += x.size
        cases.append(dict(name=name, channels=chs, sr=sr, fft=fft, window=window, noverlap=noverlap,
                          canvas_width=canvas_width, peak_mode=peak_mode, peak_threshold=peak_threshold,
                          split_channels=split, gain_db=gain, range_db=rng, scale=scale, frequency_min=fmin, frequency_max=fmax))

    sr = 384000
    add("mono_auto_overlap", [chirp(9000, sr, 61, t0=0.002)], sr, 512, noverlap=0, canvas_width=40)
    add("mono_fixed_overlap", [chirp(6000, sr, 62, t0=0.001)], sr, 512, noverlap=256, peak_threshold=0.8)
    add("stereo_split", [chirp(5000, sr, 63, t0=0.001), tone(5000, sr, 45000.0, amp=0.2, noise=1e-3, seed=64)],
        sr, 256, noverlap=128, split=True, peak_threshold=0.1)
    add("stereo_unsplit", [chirp(4000, sr, 65, t0=0.001), white(4000, 66)], sr, 256, noverlap=192, split=False,
        peak_mode=False)
    add("quiet_frames", [np.concatenate([1e-6 * white(3000, 67, 1.0), tone(3000, sr, 30000.0, amp=0.5)])], sr, 512,
        noverlap=384, peak_threshold=0.4)
    add("wide_canvas", [chirp(1150, sr, 68, t0=0.0005)], sr, 1024, noverlap=0, canvas_width=2000, peak_mode=False)
    # the non-linear frequency scales: two-tap filter banks built by the plugin, applied by the engine
    for k, scale in enumerate(("mel", "logarithmic", "bark", "erb")):
        add("scale_" + scale, [chirp(5000, sr, 70 + k, t0=0.001)], sr, 512, noverlap=256, scale=scale, peak_threshold=0.4)
    add("scale_mel_band", [chirp(5000, sr, 75, t0=0.001)], sr, 256, noverlap=128, scale="mel", fmin=10000, fmax=120000, peak_mode=False)
    add("scale_log_band_1024", [chirp(6000, sr, 76, t0=0.001)], 250000, 1024, noverlap=768, scale="logarithmic", fmin=5000, fmax=0,
        peak_mode=False)
    return cases, np.concatenate(bufs)


def main():
    """The fixture is frozen data: nothing here executes anything that ships inside the
    reference.  build_cases() remains as the record of the inputs the fixture was made from."""
    cases = build_cases()
    n = len(cases[0]) if isinstance(cases, tuple) else len(cases)
    print(f"{n} case definitions; tests/golden/golden_plugin.json is frozen and is not regenerated "
          "(its expected outputs were recorded once from the reference's own code paths; see the module docstring)")
    return 1


if __name__ == "__main__":
    sys.exit(main())
