#!/usr/bin/env python3
"""Case definitions (seeded inputs) of tests/golden/golden.npz.

TEST INFRASTRUCTURE ONLY.  The fixture holds these inputs together with the
outputs the reference's exports gave for them (windows, compute_spectrogram,
compute_spectrogram_u8, get_peaks, get_peak_magnitudes, get_global_max,
compute_power_spectrum, find_peak_frequency_from_spectrum, compute_wave_peaks:
spectrogram_wasm.d.ts:60-139,226-271).  The outputs were recorded once, in the
build container; the fixture is frozen data and is never regenerated -- nothing
in this repository loads or executes anything that ships inside the reference.
What remains here is the record of how every input was made (build_cases) and the
signal generators the tests import (chirp, tone, white).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "golden.npz")

WINDOWS = ["bartlett", "bartlettHann", "blackman", "cosine", "gauss", "hamming", "hann",
           "lanczos", "rectangular", "triangular", "no_such_window"]


def chirp(n, sr, seed, f0=80e3, f1=30e3, dur=0.005, period=0.1, amp=0.5, noise=0.002, t0=0.0005):
    """Train of downward linear-FM pulses plus uniform noise (bat-pass-like)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    tt = np.mod(t - t0, period)
    on = (tt < dur) & (t >= t0)
    phase = 2 * np.pi * (f0 * tt + 0.5 * (f1 - f0) / dur * tt * tt)
    x = np.where(on, amp * np.sin(phase), 0.0) + noise * (rng.random(n) - 0.5)
    return x.astype(np.float32)


def tone(n, sr, freq, amp=0.3, noise=0.0, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    return (amp * np.sin(2 * np.pi * freq * t) + noise * (rng.random(n) - 0.5)).astype(np.float32)


def white(n, seed, amp=0.1):
    rng = np.random.default_rng(seed)
    return (amp * rng.standard_normal(n)).astype(np.float32)


def build_cases():
    cases, bufs = [], []
    pos = [0]

    def put(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        off = pos[0]
        bufs.append(a)
        pos[0] += a.size
        return off, int(a.size)

    # --- window tables (lib.rs:698-776) ---
    for n in (256, 512, 1024):
        for w in WINDOWS:
            cases.append(dict(kind="window", name=f"win_{w}_{n}", fft=n, window=w, alpha=None))
    for n in (2048, 4096):
        for w in ("hann", "hamming", "blackman", "gauss"):
            cases.append(dict(kind="window", name=f"win_{w}_{n}", fft=n, window=w, alpha=None))
    cases.append(dict(kind="window", name="win_blackman_a0.2_512", fft=512, window="blackman", alpha=0.2))

    def stft(name, x, fft, window, noverlap, gain=20.0, rng_db=80.0, alpha=None, ratios=(0.0, 0.4),
             bank=None, num_filters=0):
        off, ln = put(x)
        c = dict(kind="stft", name=name, fft=fft, window=window, alpha=alpha, noverlap=noverlap,
                 gain=gain, range=rng_db, x_off=off, x_len=ln, ratios=list(ratios),
                 bank_off=0, bank_len=0, num_filters=0)
        if bank is not None:
            boff, bln = put(bank)
            c.update(bank_off=boff, bank_len=bln, num_filters=num_filters)
        cases.append(c)

    sr = 384000
    # headline shape: N=1024, Hann, 75 % overlap, chirp + noise (three seeds)
    for seed in (1, 2, 3):
        stft(f"chirp1024_s{seed}", chirp(1024 + 23 * 256, sr, seed, t0=0.002 * seed), 1024, "hann", 768)
    stft("noise1024_v512", white(1024 + 7 * 512, 11), 1024, "hann", 512)
    stft("tonehalf1024_hop32", tone(1024 + 15 * 32, sr, 375.0 * 100.5, noise=1e-4, seed=5), 1024, "hann", 992)
    stft("toneonbin1024_hamming", tone(1024 + 7 * 256, sr, 375.0 * 64, noise=1e-4, seed=6), 1024, "hamming", 768)
    stft("chirp1024_blackman_v0", chirp(1024 * 6, sr, 7, t0=0.001), 1024, "blackman", 0)
    stft("chirp1024_gain0_range100", chirp(1024 + 7 * 256, sr, 8, t0=0.001), 1024, "hann", 768, 0.0, 100.0)
    stft("chirp1024_gain30_range60", chirp(1024 + 7 * 256, sr, 8, t0=0.001), 1024, "hann", 768, 30.0, 60.0)
    stft("loud1024", tone(1024 + 3 * 256, sr, 40000.0, amp=1.0), 1024, "hann", 768)
    stft("silence1024", np.zeros(1024 + 3 * 256, np.float32), 1024, "hann", 768)
    stft("dc0.05_1024", np.full(1024 + 3 * 256, 0.05, np.float32), 1024, "hann", 768)
    stft("dc0.005_1024", np.full(1024 + 3 * 256, 0.005, np.float32), 1024, "hann", 768)
    imp = np.zeros(1024 + 7 * 256, np.float32)
    imp[1500] = 1.0
    stft("impulse1024", imp, 1024, "hann", 768)
    onset = np.concatenate([1e-4 * white(2000, 21, 1.0), tone(2000, sr, 50000.0, amp=0.8)]).astype(np.float32)
    stft("onset1024", onset, 1024, "hann", 768)
    # config 1 shape: N=512, Hann, 50 %
    stft("chirp512_v256", chirp(512 + 15 * 256, sr, 4, t0=0.001), 512, "hann", 256)
    # frame-count edges at N=256 (n < N, n = N, n = N+H-1, n = N+H)
    for n in (255, 256, 383, 384):
        stft(f"edge256_n{n}", tone(n, sr, 30000.0, noise=1e-3, seed=n), 256, "hann", 128)
    # every window name at N=256
    for w in WINDOWS:
        stft(f"allwin256_{w}", tone(256 + 3 * 64, 250000, 41000.0, noise=1e-2, seed=9), 256, w, 192)
    # large sizes: config 4 shape (N=4096 Blackman 95 %) and N=2048
    stft("chirp2048_v1536", chirp(2048 + 7 * 512, sr, 12, t0=0.001), 2048, "hann", 1536)
    stft("chirp4096_blackman95", chirp(4096 + 7 * 205, sr, 13, t0=0.003), 4096, "blackman", 3891, alpha=0.16)
    stft("chirp500k_1024", chirp(1024 + 11 * 256, 500000, 14, t0=0.001), 1024, "hann", 768)
    # filter bank (cold path; row stride quirk lib.rs:317-326): rows laid out with B+1 entries
    B = 128
    nf = 16
    bank = np.zeros((nf, B + 1), np.float32)
    for i in range(nf):
        c = 4 + i * 7
        bank[i, c] = 0.75
        bank[i, c + 1] = 0.25
    stft("bank256", tone(256 + 3 * 64, 250000, 41000.0, noise=1e-2, seed=10), 256, "hann", 192,
         bank=bank.ravel(), num_filters=nf)

    # --- compute_power_spectrum / find_peak_frequency (lib.rs:1019-1179) ---
    def power(name, x, sr_, fft, window, overlap, queries):
        off, ln = put(x)
        cases.append(dict(kind="power", name=name, fft=fft, window=window, sr=sr_, overlap=overlap,
                          x_off=off, x_len=ln, queries=queries))

    sel = chirp(6000, sr, 31, t0=0.0005) + np.float32(0.01)
    q = [[10000.0, 120000.0], [30000.0, 60000.0], [0.0, 192000.0], [50000.0, 40000.0]]
    for ov in (None, 0.0, 50.0, 75.0, 99.0, 100.0):
        power(f"ps1024_hann_ov{ov}", sel, sr, 1024, "hann", ov, q)
    power("ps512_blackman", sel, sr, 512, "blackman", 50.0, q)
    power("ps2048_gauss", sel, sr, 2048, "gauss", 87.5, q)
    power("ps256_Hann_upper", sel[:700], 256000, 256, "Hann", 25.0, q)   # unknown name -> hann
    power("ps1024_short", sel[:1000], sr, 1024, "hann", 50.0, q)        # n < N -> empty
    power("ps1024_tone", tone(4096, sr, 45123.0, amp=0.2, noise=1e-3, seed=3), sr, 1024, "hamming", 75.0, q)

    # --- waveform helpers (lib.rs:794-840) ---
    wv = chirp(5000, sr, 41)
    for npk in (1, 7, 100, 5000, 6000):
        off, ln = put(wv)
        cases.append(dict(kind="wave", name=f"wave_{npk}", num_peaks=npk, x_off=off, x_len=ln))
    return cases, np.concatenate(bufs)


def main():
    """The fixture is frozen data: nothing here executes anything that ships inside the
    reference.  build_cases() remains as the record of the inputs the fixture was made from."""
    cases = build_cases()
    n = len(cases[0]) if isinstance(cases, tuple) else len(cases)
    print(f"{n} case definitions; tests/golden/golden.npz is frozen and is not regenerated "
          "(its expected outputs were recorded once from the reference's own code paths; see the module docstring)")
    return 1


if __name__ == "__main__":
    sys.exit(main())
