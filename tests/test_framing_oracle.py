"""The Python restatement of the detector's call-framing stage against the golden
vectors captured from the reference's own JavaScript (tests/golden/golden_framing.npz)."""
import json
import os

import numpy as np
import pytest

from oracle import framing_oracle as fo
from oracle import sono_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(HERE, "golden", "golden_framing.npz"))
    meta = json.loads(bytes(g["__meta__"]).decode())
    return g, meta


def cases(meta, kind):
    return [c for c in meta if c["kind"] == kind]


def test_detect_calls_framing(golden):
    g, meta = golden
    seen = 0
    for c in cases(meta, "calls"):
        n = c["name"]
        e = so.OracleEngine(c["fft"], c["window"])
        noise = g[n + "/noise"] if n + "/noise" in g.files else None
        calls, zonal, floor, sp = fo.detect_calls_framing(e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"], c["hop_percent"],
                                                          c["call_threshold_db"], c["min_call_duration_ms"], noise)
        want = g[n + "/calls"].reshape(-1, 6)
        assert len(calls) == len(want), n
        for k, w in zip(calls, want):
            assert k["startTime_s"] == w[0] and k["endTime_s"] == w[1] and k["duration_ms"] == w[2], (n, k, w)   # bit-exact
            assert k["endFrame"] - k["startFrame"] + 1 == int(w[5])
            assert floor == w[3]
            peak = float(sp["powerMatrix"][k["startFrame"]:k["endFrame"] + 1].max())
            assert abs((peak - floor) - w[4]) < 2e-3                       # the dB matrix itself agrees to ~1e-4 dB
            seen += 1
        if len(want):
            assert zonal == g[n + "/zonal"].tolist(), n
    assert seen >= 8


def test_refine_end(golden):
    g, meta = golden
    for c in cases(meta, "refine"):
        got = [fo.refine_end_using_oscillogram(g[c["name"] + "/x"], c["sr"], a, b) for a, b in c["ranges"]]
        assert got == [int(v) for v in g[c["name"] + "/refined"]], c["name"]


def test_zonal_floors_percentile_and_crossings(golden):
    g, meta = golden
    for c in cases(meta, "zonal"):
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb = g[n + "/db"].reshape(F, nb), g[n + "/freq"]
        for f0, f1 in c["ranges"]:
            assert fo.zonal_noise_floors(db, fb, f0, f1) == g[f"{n}/zonal_{f0}_{f1}"].tolist(), (n, f0, f1)
        assert fo.noise_floor_percentile(db) == float(g[n + "/p25"][0])
        got = []
        for thr in c["thresholds"]:
            for f in range(0, F, c["scan_stride"]):
                for high_to_low in (True, False):
                    got.extend(fo.scan_spectrum_for_frequency(db[f], fb, thr, high_to_low))
        assert np.array_equal(np.asarray(got, np.float64), g[n + "/scans"]), n


def test_highpass_savitzky_golay_and_auto_corner(golden):
    g, meta = golden
    for c in cases(meta, "filter"):
        n = c["name"]
        x = g[n + "/x"]
        for hz, order in c["filters"]:
            got = fo.highpass_filter(x, hz, c["sr"], order)
            assert np.array_equal(got.view(np.uint32), g[f"{n}/hp_{hz}_{order}"].view(np.uint32)), (hz, order)
        assert [fo.auto_highpass_khz(p) for p in c["peaks_khz"]] == g[n + "/auto"].tolist()
        sg = fo.savitzky_golay([float(v) * 1000 for v in x[: c["sg_len"]]])
        assert sg == g[n + "/sg"].tolist()
        assert fo.savitzky_golay([1, 2, 4]) == g[n + "/sg_short"].tolist()


def test_process_full_file(golden):
    g, meta = golden
    seen = 0
    for c in cases(meta, "fullfile"):
        n = c["name"]
        e = so.OracleEngine(c["fft"], c["window"])
        calls = fo.process_full_file(e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"], c["padding_ms"],
                                     c["hop_percent"], c["call_threshold_db"], c["min_call_duration_ms"])
        want = g[n + "/calls"].reshape(-1, 5)
        assert len(calls) == len(want), (n, calls, want)
        for k, w in zip(calls, want):
            assert (k["startTime_s"], k["endTime_s"], k["duration_ms"]) == (w[0], w[1], w[2]), (n, k, w)
            assert abs(k["peakPower_dB"] + 80 - w[3]) < 2e-3 and k["frames"] == int(w[4])
            seen += 1
    assert seen >= 6


def _snr_row(r, used_noise):
    nan = float("nan")
    f = lambda v: nan if v is None else v
    return [f(r["snr_dB"]), f(r["signalPowerMean_dB"]), f(r["noisePowerMean_dB"]), r["signalCount"], r["noiseCount"],
            f(r["threshold"]), f(r["max"]), f(r["min"]), 1 if used_noise else 0]


def test_rms_snr(golden):
    g, meta = golden
    for c in cases(meta, "snr"):
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb = g[n + "/db"].reshape(F, nb), g[n + "/freq"]
        noise = g[n + "/noise_db"].reshape(-1, nb) if n + "/noise_db" in g.files else None
        rows = []
        for q in c["queries"]:
            for use in (False, True):
                if use and noise is None:
                    continue
                rows.append(_snr_row(fo.rms_snr(db, fb, q[2], q[3], q[0], q[1], c["flow_khz"], c["fhigh_khz"],
                                                noise if use else None, fb), use))
        want = g[n + "/snr"].reshape(-1, 9)
        got = np.asarray(rows, np.float64)
        assert got.shape == want.shape
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want)), n
        fin = np.isfinite(want)
        assert np.array_equal(got[:, [3, 4, 8]], want[:, [3, 4, 8]])            # counts: exact
        assert np.allclose(got[fin], want[fin], rtol=1e-12, atol=1e-12), n      # sums in another order
