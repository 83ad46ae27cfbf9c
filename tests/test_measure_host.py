"""The threshold searches of the measurement stage (host C++ through the C ABI) against the golden vectors captured
from the reference's own JavaScript methods.  No GPU needed."""
import json
import os

import numpy as np
import pytest

from spectrogram_amd import measure as ms

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(HERE, "golden", "golden_framing.npz"))
    return g, json.loads(bytes(g["__meta__"]).decode())


def _zonal_first_frames(db, fb):
    from oracle import framing_oracle as fo
    floors = fo.zonal_noise_floors(db, fb, 0, min(5, len(db) - 1))
    return {10 * z: v for z, v in enumerate(floors)}


def _num(v):
    return float("nan") if v is None else float(v)


def test_threshold_searches_match_the_reference(golden):
    g, meta = golden
    rows = threw = discards = cf = 0
    for c in [c for c in meta if c["kind"] == "thresholds"]:
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb, tf = g[n + "/db"].reshape(F, nb), g[n + "/freq"], g[n + "/time"]
        low, high = g[n + "/low"].reshape(-1, 8), g[n + "/high"].reshape(-1, 9)
        for q, wl, wh in zip(c["slices"], low, high):
            a, b = max(0, q[0]), min(F - 1, q[1])
            spec, t = db[a:b + 1], tf[a:b + 2]
            peak_frame, _ = divmod(int(np.argmax(spec)), nb)
            peak = float(spec.max())
            zonal = _zonal_first_frames(db, fb) if q[2] else None
            limit = None if q[3] < 0 else q[3]
            rows += 1
            # ---- low
            if wl[7]:
                threw += 1
                with pytest.raises(ms.ReferenceThrows):
                    ms.find_optimal_low_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, limit, zonal)
            else:
                r = ms.find_optimal_low_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, limit, zonal)
                got = [r["threshold"], _num(r["lowFreq_Hz"]), _num(r["endFreq_Hz"]), _num(r.get("lowFreqFrameIdx")),
                       _num(r.get("lowFreqBinIdx")), float(r["warning"]), _num(r.get("discard")), 0.0]
                assert np.array_equal(np.asarray(got), wl, equal_nan=True), (n, q, got, wl.tolist())        # bit-exact
                discards += bool(r.get("discard"))
            # ---- high
            if wh[8]:
                threw += 1
                with pytest.raises(ms.ReferenceThrows):
                    ms.find_optimal_high_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, zonal)
            else:
                r = ms.find_optimal_high_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, zonal)
                got = [r["threshold"], _num(r["highFreq_Hz"]), _num(r.get("highFreqBinIdx")), float(r["highFreqFrameIdx"]),
                       _num(r["startFreq_Hz"]), _num(r.get("finalSearchLimitFrame")), float(r["warning"]),
                       _num(r.get("isCFStablePattern")), 0.0]
                assert np.array_equal(np.asarray(got), wh, equal_nan=True), (n, q, got, wh.tolist())
                cf += bool(r.get("isCFStablePattern"))
    assert rows >= 35 and threw >= 1 and discards >= 1 and cf >= 1


def test_empty_matrix_defaults():
    fb = np.arange(4, dtype=np.float32) * 1000
    r = ms.find_optimal_low_frequency_threshold(np.zeros((0, 4), np.float32), [], fb, 10, 100, -30.0)
    assert r["threshold"] == -24 and r["lowFreq_Hz"] is None and r["warning"] is False
    r = ms.find_optimal_high_frequency_threshold(np.zeros((0, 4), np.float32), [], fb, 10, 100, -30.0)
    assert r["threshold"] == -1 and r["highFreq_Hz"] is None
