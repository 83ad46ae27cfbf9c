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
