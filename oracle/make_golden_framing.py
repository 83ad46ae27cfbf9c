#!/usr/bin/env python3
"""Case definitions (seeded inputs) of tests/golden/golden_framing.npz: the detector's
call-framing and measurement stages (detectCalls, processFullFile,
refineEndUsingOscillogram, calculateZonalNoiseFloors, _scanSpectrumForFrequency,
applyHighpassFilter, savitzkyGolay, measureFrequencyParameters and its threshold
searches; batCallDetector.js:584-840, 1070-1289, 1610-2909, 2994-3894, 4028-4140).

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

OUT = os.path.join(ROOT, "tests", "golden", "golden_framing.npz")


def pulse(n, sr, seed, at, dur_ms, f0, f1, amp, tail_db=None, noise=0.02):
    """Noise with one downward FM pulse starting at sample `at`; tail_db adds a weaker echo right after it."""
    x = noise * white(n, seed, 1.0)
    m = int(sr * dur_ms / 1000)
    t = np.arange(m) / sr
    ph = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / (m / sr) * t * t)
    env = np.hanning(m)
    x[at:at + m] += (amp * env * np.sin(ph)).astype(np.float32)
    if tail_db is not None:
        k = at + m + int(0.0006 * sr)
        e = amp * 10 ** (tail_db / 20)
        x[k:k + m] += (e * env * np.sin(ph)).astype(np.float32)
    return x.astype(np.float32)


def build_cases():
    cases, bufs, pos = [], [], [0]

    def put(x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        off = pos[0]
        bufs.append(x)
        pos[0] += x.size
        return off, int(x.size)

    def add(name, kind, x, sr, fft=1024, window="hann", flow=10.0, fhigh=128.0, hop_percent=3.125, call_thr=-24.0,
            min_ms=1.0, noise=None, **extra):
        off, ln = put(x)
        noff, nlen = put(noise) if noise is not None else (0, 0)
        cases.append(dict(name=name, kind=kind, fft=fft, window=window, sr=sr, flow_khz=flow, fhigh_khz=fhigh,
                          hop_percent=hop_percent, call_threshold_db=call_thr, min_call_duration_ms=min_ms,
                          x_off=off, x_len=ln, noise_off=noff, noise_len=nlen, **extra))

    sr = 384000
    # --- detectCalls framing: pulses of different lengths, an echo that the oscillogram refinement cuts, a pulse that
    #     is too short, a pulse at the very start, external noise spectrogram vs. the first-frames fallback
    a = pulse(12000, sr, 61, 3000, 4.0, 80000, 35000, 0.5)
    add("one_pulse", "calls", a, sr, flow=20.0, fhigh=100.0)
    b = pulse(16000, sr, 62, 2500, 3.0, 70000, 40000, 0.6, tail_db=-30.0)
    add("pulse_with_echo", "calls", b, sr, flow=20.0, fhigh=100.0)
    c = pulse(20000, sr, 63, 2000, 5.0, 90000, 30000, 0.4)
    c += pulse(20000, sr, 64, 11000, 0.4, 60000, 55000, 0.3, noise=0.0)        # 0.4 ms: dropped by the duration filter
    c += pulse(20000, sr, 65, 15000, 2.5, 50000, 45000, 0.5, noise=0.0)
    add("three_pulses_min2ms", "calls", c, sr, flow=20.0, fhigh=100.0, min_ms=2.0, hop_percent=6.25)
    d = pulse(9000, sr, 66, 0, 3.0, 60000, 30000, 0.5, noise=0.1)
    add("pulse_at_start_ext_noise", "calls", d, sr, flow=15.0, fhigh=110.0, noise=0.1 * white(6000, 67, 1.0), hop_percent=6.25)
    add("loud_noise_only", "calls", 0.3 * white(6000, 68, 1.0), sr, call_thr=-3.0)
    add("pulses500k_512", "calls", pulse(15000, 500000, 69, 4000, 3.5, 110000, 50000, 0.5, tail_db=-26.0), 500000, fft=512,
        flow=20.0, fhigh=150.0, hop_percent=6.25)
    # --- the whole detectCalls (measurement stage, classifier, SNR loop) on the same signals, debug mode on
    for nm, sig, kw in (("one_pulse", a, dict(flow=20.0, fhigh=100.0)), ("pulse_with_echo", b, dict(flow=20.0, fhigh=100.0)),
                        ("three_pulses_min2ms", c, dict(flow=20.0, fhigh=100.0, min_ms=2.0, hop_percent=6.25)),
                        ("pulse_at_start_ext_noise", d, dict(flow=15.0, fhigh=110.0, noise=0.1 * white(6000, 67, 1.0), hop_percent=6.25)),
                        ("loud_noise_only", 0.3 * white(6000, 68, 1.0), dict(call_thr=-3.0))):
        add("detect_" + nm, "calls_measured", sig, sr, real_measure=1, debug_mode=1, **kw)
    # --- processFullFile: fast scan -> ROIs -> auto high-pass -> segments -> strongest-first 30 ms echo filter -> framing
    n = 120000
    ff = 0.01 * white(n, 80, 1.0)
    for k, (at, dur, f0, f1, amp) in enumerate([(8000, 4.0, 80000, 45000, 0.5),       # peak >= 40 kHz: 30 kHz high-pass
                                                 (13000, 3.0, 80000, 45000, 0.05),     # echo 13 ms later: dropped (< 30 ms)
                                                 (40000, 5.0, 38000, 33000, 0.4),      # 35-40 kHz: 25 kHz high-pass
                                                 (70000, 6.0, 33000, 30500, 0.4),      # 30-35 kHz: 20 kHz
                                                 (95000, 4.0, 27000, 21000, 0.5),      # below 30 kHz: no high-pass
                                                 (110000, 0.6, 60000, 58000, 0.4)]):   # very short
        ff = ff + pulse(n, sr, 81 + k, at, dur, f0, f1, amp, noise=0.0)
    ff[52000:52400] += 0.8 * white(400, 89, 1.0)                                        # a 1 ms burst of white noise: broadband
    ff[101000:101150] += 0.9 * white(150, 88, 1.0)
    add("full_file", "fullfile", ff, sr, flow=15.0, fhigh=110.0, hop_percent=6.25, scan_threshold_db=-45.0, padding_ms=5.0)
    add("full_file_512_quiet", "fullfile", 0.001 * white(30000, 90, 1.0), 500000, fft=512, flow=20.0, fhigh=120.0,
        hop_percent=12.5, scan_threshold_db=-30.0, padding_ms=5.0)
    g2 = 0.02 * white(60000, 91, 1.0) + pulse(60000, 500000, 92, 20000, 3.0, 90000, 50000, 0.6, tail_db=-25.0, noise=0.0)
    add("full_file_512", "fullfile", g2, 500000, fft=512, flow=20.0, fhigh=120.0, hop_percent=12.5, scan_threshold_db=-50.0,
        padding_ms=8.0)
    # --- oscillogram refinement on its own: ranges that end in noise, in the signal body, out of bounds, too short
    e = pulse(14000, sr, 70, 2000, 6.0, 60000, 40000, 0.7, tail_db=-28.0)
    add("refine", "refine", e, sr, ranges=[[1500, 9000], [1500, 5000], [0, 14000], [-200, 20000], [3000, 3100], [2000, 4300],
                                           [6000, 14000], [4000, 4191], [4000, 4193]])
    q = pulse(8000, sr, 71, 1000, 4.0, 60000, 40000, 0.9, noise=1e-6)
    add("refine_quiet_floor", "refine", q, sr, ranges=[[500, 7500], [1000, 4000], [2536, 7000]])
    # --- zonal noise floors, percentile, threshold crossings
    f = pulse(9000, sr, 72, 3000, 4.0, 100000, 25000, 0.5, noise=0.05)
    add("zonal", "zonal", f, sr, flow=0.0, fhigh=192.0, hop_percent=12.5, ranges=[[0, 5], [0, 0], [10, 40], [50, 1000], [3, 2]],
        thresholds=[-60.0, -40.0, -27.5], scan_stride=3)
    add("zonal_band", "zonal", f, sr, flow=18.0, fhigh=97.0, hop_percent=25.0, ranges=[[0, 5], [2, 30]], thresholds=[-55.0],
        scan_stride=2)
    add("zonal_digital_silence", "zonal", np.zeros(5000, np.float32), sr, flow=5.0, fhigh=60.0, hop_percent=25.0,
        ranges=[[0, 5]], thresholds=[-100.0, -170.0], scan_stride=1)
    # --- the two threshold searches of the measurement stage on call slices of different kinds
    def slices(F):
        return [[0, F, 1, -1], [0, F, 0, -1], [10, F - 10, 1, -1], [20, 70, 1, 30], [0, 5, 1, -1], [40, 41, 0, -1], [0, F, 1, 10]]
    k1 = pulse(12000, sr, 101, 3000, 4.0, 80000, 35000, 0.5, noise=0.02)                      # FM sweep in moderate noise
    add("thr_fm", "thresholds", k1, sr, flow=15.0, fhigh=120.0, hop_percent=6.25, slices=slices(170))
    k2 = pulse(12000, sr, 102, 3000, 6.0, 45000, 44000, 0.5, noise=0.002)                     # near-constant frequency, clean
    add("thr_cf", "thresholds", k2, sr, flow=15.0, fhigh=120.0, hop_percent=6.25, slices=slices(170))
    k3 = pulse(12000, sr, 103, 3000, 4.0, 60000, 30000, 0.4, noise=0.05)
    k3 += pulse(12000, sr, 104, 3000, 4.0, 120000, 60000, 0.1, noise=0.0)                     # second harmonic
    k3 += 0.05 * tone(12000, sr, 9000.0, amp=1.0)                                             # hum below 10 kHz
    add("thr_harmonic_hum", "thresholds", k3, sr, flow=5.0, fhigh=150.0, hop_percent=6.25, slices=slices(170))
    k4 = pulse(9000, sr, 105, 2500, 3.0, 70000, 40000, 0.6, tail_db=-20.0, noise=1e-5)        # almost no noise: deep thresholds
    add("thr_clean_echo", "thresholds", k4, sr, flow=20.0, fhigh=100.0, hop_percent=12.5, slices=slices(60))
    k5 = pulse(15000, 500000, 106, 4000, 3.5, 110000, 50000, 0.5, noise=0.03)
    add("thr_500k_512", "thresholds", k5, 500000, fft=512, flow=20.0, fhigh=150.0, hop_percent=12.5, slices=slices(220))
    # --- the whole measureFrequencyParameters on the same kinds of slices, and inside processFullFile
    def mslices(F):
        return [[0, F, 1], [0, F, 0], [10, F - 10, 1], [20, 70, 1], [0, 5, 1], [30, 34, 1]]
    add("measure_fm", "measure", k1, sr, flow=15.0, fhigh=120.0, hop_percent=6.25, slices=mslices(170))
    add("measure_cf", "measure", k2, sr, flow=15.0, fhigh=120.0, hop_percent=6.25, slices=mslices(170))
    add("measure_harmonic_hum", "measure", k3, sr, flow=5.0, fhigh=150.0, hop_percent=6.25, slices=mslices(170))
    add("measure_clean_echo", "measure", k4, sr, flow=20.0, fhigh=100.0, hop_percent=12.5, slices=mslices(60))
    add("measure_500k_512", "measure", k5, 500000, fft=512, flow=20.0, fhigh=150.0, hop_percent=12.5, slices=mslices(220))
    k6 = 0.01 * white(14000, 107, 1.0)                                                      # hooked call: steep sweep, then nearly flat
    mm = int(sr * 0.006)
    tt = np.arange(mm) / sr
    fi = np.where(tt < 0.002, 90000 - 2.0e7 * tt, 50000 - 1.0e6 * (tt - 0.002))
    k6[3000:3000 + mm] += (0.5 * np.hanning(mm) * np.sin(2 * np.pi * np.cumsum(fi) / sr)).astype(np.float32)
    add("measure_hook", "measure", k6, sr, flow=15.0, fhigh=120.0, hop_percent=6.25, slices=mslices(200))
    add("full_file_measured", "fullfile_measured", ff, sr, flow=15.0, fhigh=110.0, hop_percent=6.25, scan_threshold_db=-45.0, padding_ms=5.0,
        real_measure=1, debug_mode=1)
    # --- RMS-based SNR of a call box, against the rest of the selection or against an external noise spectrogram
    h = pulse(9000, sr, 74, 3000, 4.0, 80000, 40000, 0.5, noise=0.03)
    add("snr", "snr", h, sr, flow=20.0, fhigh=110.0, hop_percent=12.5, noise=0.03 * white(5000, 75, 1.0),
        queries=[[40.0, 80.0, 20, 36], [45.0, 60.0, 24, 30], [20.0, 110.0, 0, 10000], [100.0, 105.0, 0, 5], [500.0, 600.0, 0, 5],
                 [50.0, 50.2, 25, 25], [50.2, 50.3, 25, 25], [60.0, 40.0, 10, 20], [40.0, 80.0, 30, 20]])
    add("snr_no_noise_case", "snr", h[:3000], sr, flow=30.0, fhigh=60.0, hop_percent=25.0, queries=[[30.0, 60.0, 0, 100], [0.0, 200.0, 0, 3]])
    # --- Butterworth high-pass, automatic corner frequency, Savitzky-Golay
    g = chirp(6000, sr, 73, period=0.01, t0=0.002) + 0.2 * tone(6000, sr, 8000.0, amp=1.0)
    add("filter", "filter", g, sr, filters=[[30000, 4], [25000, 2], [20000, 1], [20000, 3], [15000, 8], [10000, 12], [0, 4],
                                            [192000, 4], [250000, 2]],
        peaks_khz=[45.0, 40.0, 39.999, 35.0, 34.0, 30.0, 29.9, 0.0], sg_len=64)
    return cases, np.concatenate(bufs)


def main():
    """The fixture is frozen data: nothing here executes anything that ships inside the
    reference.  build_cases() remains as the record of the inputs the fixture was made from."""
    cases = build_cases()
    n = len(cases[0]) if isinstance(cases, tuple) else len(cases)
    print(f"{n} case definitions; tests/golden/golden_framing.npz is frozen and is not regenerated "
          "(its expected outputs were recorded once from the reference's own code paths; see the module docstring)")
    return 1


if __name__ == "__main__":
    sys.exit(main())
