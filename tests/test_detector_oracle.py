"""The numpy restatement of the detector front end (oracle/detector_oracle.py)
against golden vectors captured from the reference's own JavaScript
(oracle/make_golden_detector.py)."""
import json
import os

import numpy as np
import pytest

from oracle import detector_oracle as do
from oracle import sono_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_detector.npz")


def load():
    g = np.load(GOLDEN)
    return g, json.loads(bytes(g["__meta__"]).decode())


def pairs(a):
    return [tuple(int(v) for v in r) for r in np.asarray(a).reshape(-1, 2)]


G, META = load()


@pytest.mark.parametrize("c", META, ids=[c["name"] for c in META])
def test_detector_oracle_matches_reference(c):
    n = c["name"]
    x = G[n + "/x"]
    e = so.OracleEngine(c["fft"], c["window"])
    segs = do.fast_scan_segments(e, x, c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"])
    assert segs == pairs(G[n + "/scan"])
    assert do.merge_and_pad_segments(segs, len(x), c["sr"], c["padding_ms"]) == pairs(G[n + "/merged"])
    sp = do.generate_spectrogram(e, x, c["sr"], c["flow_khz"], c["fhigh_khz"], c["hop_percent"])
    F, nb = (int(v) for v in G[n + "/shape"])
    if F == 0:
        assert sp is None
        return
    assert sp["powerMatrix"].shape == (F, nb)
    ref = G[n + "/db"].reshape(F, nb)
    # 0.01 dB wherever a bin lies within 70 dB of its frame's peak (deeper bins are FFT rounding noise)
    live = ref > ref.max(axis=1, keepdims=True) - 70.0
    assert np.abs(sp["powerMatrix"] - ref)[live].max() <= 0.01
    assert np.array_equal(sp["timeFrames"], G[n + "/time"]) and np.array_equal(sp["freqBins"], G[n + "/freq"])
    assert sp["freqResolution"] == G[n + "/freqres"][0]
    # the segment rule itself is exact on the reference's own matrix
    assert do.detect_call_segments(ref, c["call_threshold_db"]) == pairs(G[n + "/calls"])
    assert do.detect_call_segments(sp["powerMatrix"], c["call_threshold_db"]) == pairs(G[n + "/calls"])
