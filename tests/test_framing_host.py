"""The host-side (serial, per-call) parts of the call-framing stage, through the
C ABI, against the golden vectors captured from the reference's JavaScript.
These entry points need no GPU."""
import json
import os

import numpy as np
import pytest

from oracle import detector_oracle as do
from oracle import sono_oracle as so
from spectrogram_amd import framing as fr

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(HERE, "golden", "golden_framing.npz"))
    return g, json.loads(bytes(g["__meta__"]).decode())


def cases(meta, kind):
    return [c for c in meta if c["kind"] == kind]


def test_refine_end_bit_exact(golden):
    g, meta = golden
    for c in cases(meta, "refine"):
        got = [fr.refine_end_using_oscillogram(g[c["name"] + "/x"], c["sr"], a, b) for a, b in c["ranges"]]
        assert got == [int(v) for v in g[c["name"] + "/refined"]], c["name"]
    with pytest.raises(Exception):
        fr.refine_end_using_oscillogram(np.zeros(100, np.float32), 8000, 0, 100)       # 0.1 ms window has no hop


def test_refine_end_against_the_oracle_on_awkward_spans():
    """The product sums the 0.1 ms windows several at a time and evaluates the dB only where the walk needs it; the
    restatement does what the reference does, window after window.  Same answer on: decaying tone bursts, spans whose
    windows have EXACTLY equal energies (the first of them is the peak), spans below -60 dB, silence, denormals, and
    samples that are NaN or infinite."""
    from oracle import framing_oracle as fo
    sr = 384000
    rng = np.random.default_rng(11)
    n = 12000
    t = np.arange(n) / sr
    burst = (np.sin(2 * np.pi * 45000 * t) * np.exp(-np.maximum(t - 0.008, 0) * 2500) * (t > 0.004)).astype(np.float32)
    square = np.where((np.arange(n) // 19) % 2 == 0, 0.5, -0.5).astype(np.float32)             # every window: 38 x 0.25
    square_then_quiet = square.copy()
    square_then_quiet[6000:] = 1e-5 * rng.standard_normal(n - 6000).astype(np.float32)
    quiet = (2e-4 * rng.standard_normal(n)).astype(np.float32)                                   # loudest window below -60 dB
    quiet_flat = np.full(n, 1e-4, np.float32)                                                    # equal energies, quiet
    tiny = np.full(n, 1e-30, np.float32)
    denormal = np.full(n, 1e-42, np.float32)
    noisy_burst = (0.7 * burst + 1e-3 * rng.standard_normal(n)).astype(np.float32)
    with_nan = noisy_burst.copy(); with_nan[5000] = np.nan
    with_inf = noisy_burst.copy(); with_inf[7000] = np.inf
    two_equal_peaks = np.zeros(n, np.float32); two_equal_peaks[2000:2038] = 0.5; two_equal_peaks[4000:4038] = 0.5
    signals = {"burst": burst, "square": square, "square_then_quiet": square_then_quiet, "quiet": quiet, "quiet_flat": quiet_flat,
               "zeros": np.zeros(n, np.float32), "tiny": tiny, "denormal": denormal, "noisy_burst": noisy_burst, "with_nan": with_nan,
               "with_inf": with_inf, "two_equal_peaks": two_equal_peaks}
    ranges = [(0, n), (1000, 9000), (1536, 6100), (-50, n + 50), (3000, 3100), (3000, 3300), (1999, 4100), (1981, 4057)]
    ranges += [(int(a), int(a + b)) for a, b in zip(rng.integers(0, n - 400, 12), rng.integers(200, 9000, 12))]
    moved = 0
    for name, x in signals.items():
        for a, b in ranges:
            want = fo.refine_end_using_oscillogram(x, sr, a, b)
            got = fr.refine_end_using_oscillogram(x, sr, a, b)
            assert got == want, (name, a, b, got, want)
            moved += want != b
    assert moved >= 20


def test_frame_calls_bit_exact(golden):
    # segments and time stamps from the CPU oracle's spectrogram, framing through the C ABI
    g, meta = golden
    seen = 0
    for c in cases(meta, "calls"):
        n = c["name"]
        x = g[n + "/x"]
        sp = do.generate_spectrogram(so.OracleEngine(c["fft"], c["window"]), x, c["sr"], c["flow_khz"], c["fhigh_khz"],
                                     c["hop_percent"])
        segs = [{"startFrame": a, "endFrame": b} for a, b in do.detect_call_segments(sp["powerMatrix"], c["call_threshold_db"])]
        calls = fr.frame_calls(segs, sp["timeFrames"], x, c["sr"], c["fft"], c["min_call_duration_ms"])
        want = g[n + "/calls"].reshape(-1, 6)
        assert len(calls) == len(want), n
        for k, w in zip(calls, want):
            assert (k["startTime_s"], k["endTime_s"], k["duration_ms"]) == (w[0], w[1], w[2]), (n, k, w)
            assert k["endFrame"] - k["startFrame"] + 1 == int(w[5])
            seen += 1
    assert seen >= 8
    assert fr.frame_calls([], [0.0, 1.0], np.zeros(10, np.float32), 384000, 1024) == []


def test_threshold_crossings(golden):
    g, meta = golden
    for c in cases(meta, "zonal"):
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb = g[n + "/db"].reshape(F, nb), g[n + "/freq"]
        got = []
        for thr in c["thresholds"]:
            for f in range(0, F, c["scan_stride"]):
                for d in ("HighToLow", "LowToHigh"):
                    r = fr.scan_spectrum_for_frequency(db[f], fb, thr, d)
                    got.extend([r["freq_Hz"], r["binIdx"]] if r else [-1, -1])
        assert np.array_equal(np.asarray(got, np.float64), g[n + "/scans"]), n


def test_highpass_bit_exact_and_small_helpers(golden):
    g, meta = golden
    for c in cases(meta, "filter"):
        n = c["name"]
        x = g[n + "/x"]
        for hz, order in c["filters"]:
            got = fr.apply_highpass_filter(x, hz, c["sr"], order)
            assert np.array_equal(got.view(np.uint32), g[f"{n}/hp_{hz}_{order}"].view(np.uint32)), (hz, order)
        assert [fr.calculate_auto_highpass_filter_freq(p) for p in c["peaks_khz"]] == g[n + "/auto"].tolist()
        assert fr.savitzky_golay([float(v) * 1000 for v in x[: c["sg_len"]]]) == g[n + "/sg"].tolist()
        assert fr.savitzky_golay([1, 2, 4]) == g[n + "/sg_short"].tolist()
    assert len(fr.apply_highpass_filter(np.zeros(0, np.float32), 1000, 384000)) == 0
