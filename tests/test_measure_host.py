"""The measurement stage (host C++ through the C ABI) against the golden vectors captured from the reference's own
JavaScript methods: both threshold searches, inside and outside the detector's debug mode, and the whole
measureFrequencyParameters.  No GPU needed."""
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


@pytest.mark.parametrize("debug", [False, True])
def test_threshold_searches_match_the_reference(golden, debug):
    g, meta = golden
    rows = threw = discards = cf = 0
    sfx = "_debug" if debug else ""
    for c in [c for c in meta if c["kind"] == "thresholds"]:
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb, tf = g[n + "/db"].reshape(F, nb), g[n + "/freq"], g[n + "/time"]
        low, high = g[n + "/low" + sfx].reshape(-1, 8), g[n + "/high" + sfx].reshape(-1, 9)
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
                    ms.find_optimal_low_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, limit, zonal, debug)
            else:
                r = ms.find_optimal_low_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, limit, zonal, debug)
                got = [r["threshold"], _num(r["lowFreq_Hz"]), _num(r["endFreq_Hz"]), _num(r.get("lowFreqFrameIdx")),
                       _num(r.get("lowFreqBinIdx")), float(r["warning"]), _num(r.get("discard")), 0.0]
                assert np.array_equal(np.asarray(got), wl, equal_nan=True), (n, q, got, wl.tolist())        # bit-exact
                discards += bool(r.get("discard"))
            # ---- high
            if wh[8]:
                threw += 1
                with pytest.raises(ms.ReferenceThrows):
                    ms.find_optimal_high_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, zonal, debug)
            else:
                r = ms.find_optimal_high_frequency_threshold(spec, t, fb, c["flow_khz"], c["fhigh_khz"], peak, peak_frame, zonal, debug)
                got = [r["threshold"], _num(r["highFreq_Hz"]), _num(r.get("highFreqBinIdx")), float(r["highFreqFrameIdx"]),
                       _num(r["startFreq_Hz"]), _num(r.get("finalSearchLimitFrame")), float(r["warning"]),
                       _num(r.get("isCFStablePattern")), 0.0]
                assert np.array_equal(np.asarray(got), wh, equal_nan=True), (n, q, got, wh.tolist())
                cf += bool(r.get("isCFStablePattern"))
    assert rows >= 35 and discards >= 1 and cf >= 1
    assert threw == 0 if debug else threw >= 1


# columns of the golden rows: the BatCall fields the capture script lists, then isDiscarded and "threw"
_GOLDEN_FIELDS = ["startTime_s", "endTime_s", "duration_ms", "peakFreq_kHz", "peakFreq_ms", "peakPower_dB", "highFreq_kHz", "highFreq_ms",
                  "highFreqFrameIdx", "startFreq_kHz", "startFreq_ms", "startFreqFrameIdx", "endFreq_kHz", "endFreq_ms",
                  "endFrameIdx_forLowFreq", "lowFreq_kHz", "lowFreq_ms", "characteristicFreq_kHz", "characteristicFreq_ms",
                  "kneeFreq_kHz", "kneeFreq_ms", "kneeFrameIdx", "heelFreq_kHz", "heelFreq_ms", "heelFrameIdx", "bandwidth_kHz",
                  "highFreqThreshold_dB_used", "lowFreqThreshold_dB_used", "snr_dB", "startFreqTime_s", "endFreqTime_s"]


def _row_of(call):
    return np.asarray([_num(call.get(k)) for k in _GOLDEN_FIELDS])


@pytest.mark.parametrize("debug", [False, True])
def test_whole_measurement_matches_the_reference(golden, debug):
    g, meta = golden
    measured = discarded = knees = heels = 0
    for c in [c for c in meta if c["kind"] == "measure"]:
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb, tf = g[n + "/db"].reshape(F, nb), g[n + "/freq"], g[n + "/time"]
        res = float(g[n + "/freqres"][0])
        want = g[n + ("/measured_debug" if debug else "/measured")].reshape(-1, len(_GOLDEN_FIELDS) + 2)
        for q, w in zip(c["slices"], want):
            a, b = max(0, q[0]), min(F - 1, q[1])
            call = {"spectrogram": db[a:b + 1], "timeFrames": tf[a:b + 2], "startTime_s": float(tf[a]),
                    "endTime_s": float(tf[min(b + 1, F - 1)])}
            call["duration_ms"] = (call["endTime_s"] - call["startTime_s"]) * 1000                # calculateDuration, js:158-159
            zonal = _zonal_first_frames(db, fb) if q[2] else None
            if w[-1]:
                with pytest.raises(ms.ReferenceThrows):
                    ms.measure_frequency_parameters(call, c["flow_khz"], c["fhigh_khz"], fb, res, zonal, debug)
                continue
            ms.measure_frequency_parameters(call, c["flow_khz"], c["fhigh_khz"], fb, res, zonal, debug)
            got = _row_of(call)
            # curvature only ranks contour points, so every number is an exact function of the inputs: bit-exact
            assert np.array_equal(got, w[:-2], equal_nan=True), (n, q, dict(zip(_GOLDEN_FIELDS, zip(got.tolist(), w.tolist()))))
            assert bool(call.get("isDiscarded")) == bool(w[-2])
            measured += 1
            discarded += bool(w[-2])
            knees += call["kneeFreq_kHz"] is not None
            heels += call["heelFreq_kHz"] is not None
    assert measured >= 30 and discarded >= 5 and knees >= 3, (measured, discarded, knees, heels)


_TIMES_10 = ("peakFreq_kHz", "highFreq_kHz", "startFreq_kHz", "endFreq_kHz", "lowFreq_kHz", "characteristicFreq_kHz", "kneeFreq_kHz",
             "heelFreq_kHz", "bandwidth_kHz")
_OVER_10 = ("startTime_s", "endTime_s", "duration_ms", "peakFreq_ms", "highFreq_ms", "startFreq_ms", "endFreq_ms", "lowFreq_ms",
            "characteristicFreq_ms", "kneeFreq_ms", "heelFreq_ms", "startFreqTime_s", "endFreqTime_s")


def test_time_expansion_rescales_a_measured_call(golden):
    """getTimeExpansionMode() on: BatCall.applyTimeExpansion(10) at the end of measureFrequencyParameters
    (batCallDetector.js:185-278, 3890-3893).  The fixtures were captured with the switch off, so the expected values are
    the captured ones put through the reference's rule: frequency fields * 10, time fields / 10 (one IEEE operation
    each, hence bit-exact), everything else and every call that leaves the measurement early (discarded) unchanged."""
    g, meta = golden
    scaled = untouched = 0
    for c in [c for c in meta if c["kind"] == "measure"]:
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb, tf = g[n + "/db"].reshape(F, nb), g[n + "/freq"], g[n + "/time"]
        res = float(g[n + "/freqres"][0])
        want = g[n + "/measured_debug"].reshape(-1, len(_GOLDEN_FIELDS) + 2)
        for q, w in zip(c["slices"], want):
            a, b = max(0, q[0]), min(F - 1, q[1])
            call = {"spectrogram": db[a:b + 1], "timeFrames": tf[a:b + 2], "startTime_s": float(tf[a]),
                    "endTime_s": float(tf[min(b + 1, F - 1)])}
            call["duration_ms"] = (call["endTime_s"] - call["startTime_s"]) * 1000
            zonal = _zonal_first_frames(db, fb) if q[2] else None
            ms.measure_frequency_parameters(call, c["flow_khz"], c["fhigh_khz"], fb, res, zonal, True, time_expansion=True)
            plain = dict(zip(_GOLDEN_FIELDS, w[:-2].tolist()))
            early = bool(w[-2])                        # isDiscarded: the reference returns before the correction
            for k in _GOLDEN_FIELDS:
                if k == "snr_dB":
                    continue
                e = plain[k]
                if not early:
                    e = e * 10 if k in _TIMES_10 else e / 10 if k in _OVER_10 else e
                got = _num(call.get(k))
                assert got == e or (got != got and e != e), (n, q, k, got, e)
            scaled += not early
            untouched += early
    assert scaled >= 20 and untouched >= 5, (scaled, untouched)


def test_measurement_of_an_empty_call_changes_nothing():
    fb = np.arange(4, dtype=np.float32) * 1000
    call = {"spectrogram": np.zeros((0, 4), np.float32), "timeFrames": [], "startTime_s": 1.0, "endTime_s": 2.0, "duration_ms": 1000.0}
    ms.measure_frequency_parameters(call, 10, 100, fb, 1000.0)
    assert call["startTime_s"] == 1.0 and call["peakFreq_kHz"] is None and "isDiscarded" not in call


def test_empty_matrix_defaults():
    fb = np.arange(4, dtype=np.float32) * 1000
    r = ms.find_optimal_low_frequency_threshold(np.zeros((0, 4), np.float32), [], fb, 10, 100, -30.0)
    assert r["threshold"] == -24 and r["lowFreq_Hz"] is None and r["warning"] is False
    r = ms.find_optimal_high_frequency_threshold(np.zeros((0, 4), np.float32), [], fb, 10, 100, -30.0)
    assert r["threshold"] == -1 and r["highFreq_Hz"] is None


# per field: how far a value may move when the dB matrix differs from the reference's by ~1e-4 dB (numpy vs wasm FFT);
# frame-quantised times and thresholds must not move at all
_TOL = {"peakFreq_kHz": 2e-3, "peakPower_dB": 2e-3, "highFreq_kHz": 5e-3, "startFreq_kHz": 5e-3, "endFreq_kHz": 5e-3, "lowFreq_kHz": 5e-3,
        "bandwidth_kHz": 1e-2, "snr_dB": 2e-3, "characteristicFreq_kHz": 5e-3}


def test_full_file_with_the_measurement_stage(golden):
    """The oracle's framing of a recording + the host measurement and follow-up of the product, against the reference's
    whole processFullFile (debug mode on: outside it the reference raises on one call of this recording)."""
    from oracle import framing_oracle as fo, sono_oracle as so
    g, meta = golden
    (c,) = [c for c in meta if c["kind"] == "fullfile_measured"]
    n = c["name"]
    e = so.OracleEngine(c["fft"], c["window"])
    framed = fo.process_full_file(e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"], c["padding_ms"],
                                  c["hop_percent"], c["call_threshold_db"], c["min_call_duration_ms"], keep_slices=True)
    calls, raised = [], 0
    for k in framed:
        for debug in (False, True):
            call = {"spectrogram": k["spectrogram"], "timeFrames": k["timeFrames"], "startTime_s": k["relStart_s"],
                    "endTime_s": k["relEnd_s"], "duration_ms": k["duration_ms"]}
            try:
                ms.measure_frequency_parameters(call, c["flow_khz"], c["fhigh_khz"], k["freqBins"], k["freqResolution"],
                                                k["zonalNoiseMap"], debug)
            except ms.ReferenceThrows:
                assert not debug
                raised += 1
        call = ms.finish_full_file_call(call, k["offset_s"])
        if call is not None:
            calls.append(call)
    assert raised >= 1                                             # why the capture ran in debug mode
    want = g[n + "/measured"].reshape(-1, len(_GOLDEN_FIELDS))
    assert len(calls) == len(want)
    for call, w in zip(calls, want):
        for name, got, ref in zip(_GOLDEN_FIELDS, _row_of(call), w):
            if ref != ref:
                assert got != got, (name, got)
            else:
                assert abs(got - ref) <= _TOL.get(name, 1e-9), (name, got, ref)
    assert [call["callType"] for call in calls].count("FM") >= 1 and all(call["quality"] for call in calls)


def test_measurement_rejects_bad_arguments():
    import ctypes as C
    import spectrogram_amd as sa
    from spectrogram_amd import CallMatrix, CallMeasurement
    L = sa.lib()
    db = np.zeros((4, 8), np.float32)
    fb = (np.arange(8) * 1000).astype(np.float32)
    tf = np.arange(5) * 1e-3
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    m = CallMatrix(db.ctypes.data_as(fp), 4, 8, fb.ctypes.data_as(fp), tf.ctypes.data_as(dp), None, 0, 0, 0)
    r = CallMeasurement()
    assert L.sono_measure_frequency_parameters(C.byref(m), 5, 100.0, 1000.0, None) == sa.SONO_EINVAL
    assert L.sono_measure_frequency_parameters(None, 5, 100.0, 1000.0, C.byref(r)) == sa.SONO_EINVAL
    assert L.sono_measure_frequency_parameters(C.byref(m), 3, 100.0, 1000.0, C.byref(r)) == sa.SONO_EINVAL       # fewer times than frames
    assert b"fewer times" in L.sono_last_error()
    assert L.sono_measure_frequency_parameters(C.byref(m), 5, 100.0, 0.0, C.byref(r)) == sa.SONO_EINVAL
    m.time_frames = None
    assert L.sono_measure_frequency_parameters(C.byref(m), 5, 100.0, 1000.0, C.byref(r)) == sa.SONO_EINVAL
    # the whole-file entry points check their handles before anything touches a GPU
    assert L.sono_process_full_file(None, None, 0, 384000.0, 10.0, 100.0, None, None, 0, None) == sa.SONO_EINVAL
    assert L.sono_full_file_results(None, None, 0, None) == sa.SONO_EINVAL
    assert L.sono_measure_roi_calls(None, None, None, 0, None, 0, None, None, 0, 384000.0, 10.0, 100.0, 3.125, 0, 0, None) == sa.SONO_EINVAL


def test_flat_call_shows_the_smoothing_edge_quirk():
    # Every bin equal.  The reference's Savitzky-Golay divides by the full coefficient sum at the edges too
    # (batCallDetector.js:1625-1636), so a constant contour comes out as 14.86, 21.71, 20, 20, ... and the steep fake
    # drop-and-recover at the start is taken for a knee at contour point 2.
    db = np.full((12, 16), -50.0, np.float32)
    fb = (20000 + np.arange(16) * 1000).astype(np.float32)
    tf = (np.arange(13) * 32 + 512) / 384000
    call = {"spectrogram": db, "timeFrames": tf, "startTime_s": float(tf[0]), "endTime_s": float(tf[12]), "duration_ms": 1.0}
    ms.measure_frequency_parameters(call, 20, 36, fb, 1000.0, None, True)
    assert call["peakPower_dB"] == -50.0 and call["peakFreq_kHz"] == 20.0
    assert call["kneeFrameIdx"] == 2 and call["kneeFreq_kHz"] == 20.0 and call["heelFreq_kHz"] is None
    assert call["bandwidth_kHz"] == 15.0 and call["startFreq_ms"] == 0.0 and call["endFrameIdx_forLowFreq"] == 11
