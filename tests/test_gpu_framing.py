"""The call-framing stage with its matrix passes on the GPU, against the golden
vectors captured from the reference's JavaScript.  Needs an MI355X: -m gpu."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import spectrogram_amd as sa
from oracle import framing_oracle as fo
from spectrogram_amd import detector as det
from spectrogram_amd import framing as fr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(HERE, "golden", "golden_framing.npz"))
    return g, json.loads(bytes(g["__meta__"]).decode())


def cases(meta, kind):
    return [c for c in meta if c["kind"] == kind]


def test_zonal_floors_and_percentile_on_the_reference_matrix(golden):
    # the reference's own dB matrix in, so every count and the order statistic must match exactly
    g, meta = golden
    e = sa.SpectrogramEngine(1024, "hann")
    for c in cases(meta, "zonal"):
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb = g[n + "/db"].reshape(F, nb), g[n + "/freq"]
        for f0, f1 in c["ranges"]:
            got = fr.calculate_zonal_noise_floors(e, db, fb, f0, f1)
            assert [got[k] for k in sorted(got)] == g[f"{n}/zonal_{f0}_{f1}"].tolist(), (n, f0, f1)
        assert fr.db_percentile(e, db, 0.25) == float(g[n + "/p25"][0])
        assert fr.db_percentile(e, db, 0.0) == float(db.min()) and fr.db_percentile(e, db, 1.0) == float(db.max())
    with pytest.raises(sa.SonoError):
        fr.db_percentile(e, np.zeros(0, np.float32))
    e.free()


def test_zonal_histogram_at_scale_matches_numpy():
    rng = np.random.default_rng(5)
    F, nb = 3000, 400
    db = (rng.normal(-85, 6, (F, nb)) + np.linspace(0, 25, nb)[None, :]).astype(np.float32)
    db[5, 7], db[6, 8], db[7, 9] = -np.inf, np.inf, np.nan
    fb = (np.arange(nb) * 375.0 + 10000).astype(np.float32)
    e = sa.SpectrogramEngine(1024, "hann")
    got = fr.calculate_zonal_noise_floors(e, db, fb, 10, 2900)
    want = fo.zonal_noise_floors(np.nan_to_num(db, nan=-1e9)[:, :], fb, 10, 2900)      # the NaN sits outside the frame range
    assert [got[k] for k in sorted(got)] == want
    assert fr.db_percentile(e, db[:5], 0.25) == float(np.sort(db[:5].ravel())[int(5 * nb * 0.25)])
    e.free()


def test_detect_calls_framing_end_to_end(golden):
    g, meta = golden
    seen = 0
    for c in cases(meta, "calls"):
        n = c["name"]
        e = sa.SpectrogramEngine(c["fft"], c["window"])
        cfg = {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"],
               "minCallDuration_ms": c["min_call_duration_ms"]}
        ns = None
        if n + "/noise" in g.files:
            ns = det.generate_spectrogram(e, g[n + "/noise"], c["sr"], c["flow_khz"], c["fhigh_khz"], c["hop_percent"])
        calls, zonal, floor, sp = fr.detect_calls_framing(e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"], cfg, ns)
        want = g[n + "/calls"].reshape(-1, 6)
        assert len(calls) == len(want), n
        for k, w in zip(calls, want):
            assert (k["startTime_s"], k["endTime_s"], k["duration_ms"]) == (w[0], w[1], w[2]), (n, k, w)     # call boundaries: exact
            assert abs(k["noiseFloor_dB"] - w[3]) < 1e-3
            peak = float(sp["powerMatrix"][k["startFrame"]:k["endFrame"] + 1].max())
            assert abs((peak - floor) - w[4]) < 2e-3
            seen += 1
        if len(want):
            # a dB value within 1e-4 of an integer may fall into the neighbouring 1 dB cell: allow one cell per zone
            ref = g[n + "/zonal"]
            got = np.asarray([zonal[k] for k in sorted(zonal)])
            assert got.shape == ref.shape and np.all(np.abs(got - ref) <= 1.0), (n, got, ref)
            assert np.mean(got == ref) >= 0.8
        e.free()
    assert seen >= 8


def test_process_full_file_end_to_end(golden):
    from spectrogram_amd import sharding as sh
    g, meta = golden
    seen = 0
    for c in cases(meta, "fullfile"):
        n = c["name"]
        e = sa.SpectrogramEngine(c["fft"], c["window"])
        calls = fr.process_full_file(e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"],
                                     {"threshold_dB": c["scan_threshold_db"], "padding_ms": c["padding_ms"]},
                                     {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"],
                                      "minCallDuration_ms": c["min_call_duration_ms"]})
        want = g[n + "/calls"].reshape(-1, 5)
        assert len(calls) == len(want), (n, calls, want)
        for k, w in zip(calls, want):
            assert (k["startTime_s"], k["endTime_s"], k["duration_ms"]) == (w[0], w[1], w[2]), (n, k, w)     # exact
            assert abs(k["snr_dB"] - w[3]) < 2e-3 and k["endFrame"] - k["startFrame"] + 1 == int(w[4])
            seen += 1
        if n == "full_file":
            assert sorted(k["highpass_kHz"] for k in calls) == [0, 20, 25, 30, 30]
            assert g[n + "/discarded"].sum() == 1 and all(k["instBandwidth_kHz"] <= 20.0 for k in calls)     # the noise burst is gone
            rows = sh.calls_to_rows(calls, file_id=3)
            col = {k: i for i, k in enumerate(sh.COLUMNS)}
            assert rows.shape == (len(calls), len(sh.COLUMNS)) and (rows[:, col["file"]] == 3).all()
            assert np.array_equal(rows[:, col["startTime_s"]], want[:, 0]) and np.array_equal(rows[:, col["snr_dB"]],
                                                                                             [k["snr_dB"] for k in calls])
        e.free()
    assert seen >= 6


def _same_calls(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for p, q in zip(a, b):
        for k in ("roi", "startFrame", "endFrame", "startTime_s", "endTime_s", "duration_ms", "peakPower_dB", "peakTime_s",
                  "peakFreq_kHz", "peakFreq_ms", "instBandwidth_kHz", "snr_dB", "highpass_kHz"):
            assert p[k] == q[k], (k, p, q)
        assert {int(k): v for k, v in p["zonalNoiseMap"].items()} == {int(k): v for k, v in q["zonalNoiseMap"].items()}


def test_batched_region_analysis_equals_the_region_by_region_form(golden):
    g, meta = golden
    for c in cases(meta, "fullfile"):
        n = c["name"]
        e = sa.SpectrogramEngine(c["fft"], c["window"])
        args = (e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"], {"threshold_dB": c["scan_threshold_db"], "padding_ms": c["padding_ms"]},
                {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"], "minCallDuration_ms": c["min_call_duration_ms"]})
        one, batched = fr.process_full_file(*args), fr.process_full_file_batched(*args)
        _same_calls(one, batched)
        want = g[n + "/calls"].reshape(-1, 5)
        assert [(k["startTime_s"], k["endTime_s"], k["duration_ms"]) for k in batched] == [(w[0], w[1], w[2]) for w in want]
        # the reference's own peak step, run inside the golden capture: peak frequency (parabolic), peak time, peak frame
        for k, w in zip(batched, g[n + "/peaks"].reshape(-1, 4)):
            assert abs(k["peakFreq_kHz"] - w[0]) < 2e-3                      # dB values agree to ~1e-4 dB; the parabola amplifies that
            assert k["peakFreq_ms"] == w[1] and abs(k["peakPower_dB"] - w[3]) < 2e-3
        e.free()


_MEASURED = ["startTime_s", "endTime_s", "duration_ms", "peakFreq_kHz", "peakFreq_ms", "peakPower_dB", "highFreq_kHz", "highFreq_ms",
             "highFreqFrameIdx", "startFreq_kHz", "startFreq_ms", "startFreqFrameIdx", "endFreq_kHz", "endFreq_ms",
             "endFrameIdx_forLowFreq", "lowFreq_kHz", "lowFreq_ms", "characteristicFreq_kHz", "characteristicFreq_ms",
             "kneeFreq_kHz", "kneeFreq_ms", "kneeFrameIdx", "heelFreq_kHz", "heelFreq_ms", "heelFrameIdx", "bandwidth_kHz",
             "highFreqThreshold_dB_used", "lowFreqThreshold_dB_used", "snr_dB", "startFreqTime_s", "endFreqTime_s"]
# what ~1e-4 dB between this STFT and the reference's may move; frame-quantised times and thresholds must not move
_TOL = {"peakFreq_kHz": 2e-3, "peakPower_dB": 2e-3, "highFreq_kHz": 5e-3, "startFreq_kHz": 5e-3, "endFreq_kHz": 5e-3, "lowFreq_kHz": 5e-3,
        "bandwidth_kHz": 1e-2, "snr_dB": 2e-3, "characteristicFreq_kHz": 5e-3}


def test_whole_detect_calls_matches_the_reference(golden):
    g, meta = golden
    types, qualities = ["CF", "FM", "CF-FM"], ["Very Poor", "Poor", "Normal", "Good", "Excellent"]
    seen = discarded = 0
    for c in cases(meta, "calls_measured"):
        n = c["name"]
        e = sa.SpectrogramEngine(c["fft"], c["window"])
        x = g[n + "/x"]
        noise = None
        if n + "/noise" in g.files:
            noise = det.generate_spectrogram(e, g[n + "/noise"], c["sr"], c["flow_khz"], c["fhigh_khz"], c["hop_percent"])
        cfg = {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"], "minCallDuration_ms": c["min_call_duration_ms"],
               "debugMode": True}
        for skip, key in ((False, "measured"), (True, "measured_skip_snr")):
            calls = fr.detect_calls(e, x, c["sr"], c["flow_khz"], c["fhigh_khz"], {"skipSNR": skip, "noiseSpectrogram": noise}, cfg)
            want = g[n + "/" + key].reshape(-1, len(_MEASURED) + 6)
            assert len(calls) == len(want), (n, key, len(calls), len(want))
            for k, w in zip(calls, want):
                for name, ref in zip(_MEASURED, w):
                    got = k.get(name)
                    if ref != ref:
                        assert got is None, (n, name, got)
                    else:
                        assert abs(got - ref) <= _TOL.get(name, 1e-9), (n, key, name, got, ref)
                assert abs(k["noiseFloor_dB"] - w[31]) < 2e-3 and bool(k.get("isDiscarded")) == bool(w[32])
                assert k["callType"] == types[int(w[33])] and k["quality"] == qualities[int(w[34])]
                assert abs(k["Flow"] - w[35]) <= 5.0 and (k["Fhigh"] is None if w[36] != w[36] else abs(k["Fhigh"] - w[36]) <= 5e-3)
                seen += 1
                discarded += bool(w[32])
        e.free()
    assert seen >= 14 and discarded >= 2


def test_full_file_with_the_measurement_stage(golden):
    from spectrogram_amd import measure as ms, sharding as sh
    g, meta = golden
    (c,) = cases(meta, "fullfile_measured")
    n = c["name"]
    e = sa.SpectrogramEngine(c["fft"], c["window"])
    args = (e, g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"], {"threshold_dB": c["scan_threshold_db"], "padding_ms": c["padding_ms"]})
    cfg = {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"], "minCallDuration_ms": c["min_call_duration_ms"]}
    for f in (fr.process_full_file, fr.process_full_file_batched):
        with pytest.raises(ms.ReferenceThrows):               # outside its debug mode the reference raises on this recording
            f(*args, cfg, measure=True)
    one = fr.process_full_file(*args, dict(cfg, debugMode=True), measure=True)
    batched = fr.process_full_file_batched(*args, dict(cfg, debugMode=True), measure=True)
    want = g[n + "/measured"].reshape(-1, len(_MEASURED))
    assert len(one) == len(batched) == len(want)
    for a, b, w in zip(one, batched, want):
        assert {k: v for k, v in a.items() if k != "zonalNoiseMap"} == {k: v for k, v in b.items() if k != "zonalNoiseMap"}   # bit for bit
        for name, ref in zip(_MEASURED, w):
            got = a.get(name)
            if ref != ref:
                assert got is None, (name, got)
            else:
                assert abs(got - ref) <= _TOL.get(name, 1e-9), (name, got, ref)
    assert {k["callType"] for k in one} <= {"CF", "FM", "CF-FM"} and all(k["quality"] for k in one)
    rows = sh.calls_to_rows(batched, file_id=1)
    assert rows.shape[0] == len(batched)
    # the same as ONE library call
    with pytest.raises(ms.ReferenceThrows):
        fr.process_full_file_native(*args, cfg)
    native = fr.process_full_file_native(*args, dict(cfg, debugMode=True))
    assert len(native) == len(batched)
    for a, b in zip(native, batched):
        assert a == {k: v for k, v in b.items() if k != "zonalNoiseMap"}
    # ... and straight into the call-summary table's rows
    table = fr.process_full_file_table(*args, dict(cfg, debugMode=True), file_id=1)
    assert np.array_equal(table, rows, equal_nan=True)
    # call.spectrogram / timeFrames / freqBins (js:785-787): the native run's re-transform of every call's frames
    # against the rows of the region's own matrix that the region-by-region form measured
    with_rows = fr.process_full_file_native(*args, dict(cfg, debugMode=True), keep_spectrograms=True)
    regions = fr.process_full_file(*args, dict(cfg, debugMode=True), measure=True, keep_spectrograms=True)
    assert len(with_rows) == len(regions) == len(native)
    for a, b, plain in zip(with_rows, regions, native):
        f = a["endFrame"] - a["startFrame"] + 1
        assert a["spectrogram"].shape == (f, len(a["freqBins"])) and len(a["timeFrames"]) in (f, f + 1)
        assert np.array_equal(a["spectrogram"], np.asarray(b["spectrogram"]))                       # bit for bit
        assert np.array_equal(a["timeFrames"], np.asarray(b["timeFrames"], np.float64))
        assert np.array_equal(a["freqBins"], b["freqBins"])
        assert {k: v for k, v in a.items() if k not in ("spectrogram", "timeFrames", "freqBins")} == plain
    fr.process_full_file_native(*args, dict(cfg, debugMode=True))                                   # this run keeps none
    assert sa.lib().sono_full_file_call_spectrogram(e._handle(), 0, None, 0, None, 0, None, 0, None, None, None) == sa.SONO_EINVAL
    e.free()


def test_batched_region_analysis_on_a_long_recording():
    # 6 s at 384 kHz with a pulse every 50 ms: more than a hundred regions in one batch
    sr = 384000
    rng = np.random.default_rng(12)
    x = (0.01 * rng.standard_normal(6 * sr)).astype(np.float32)
    t = np.arange(1536) / sr
    for k in range(0, 6 * sr - 4000, 19200):
        f0 = [80000, 38000, 33000, 25000][(k // 19200) % 4]          # one pulse for each class of the automatic high-pass
        x[k + 500:k + 2036] += (0.4 * np.hanning(1536) * np.sin(2 * np.pi * (f0 * t - 1e5 * t * t))).astype(np.float32)
    e = sa.SpectrogramEngine(1024, "hann")
    cfg = {"hopPercent": 6.25}
    one = fr.process_full_file(e, x, sr, 15.0, 110.0, {"threshold_dB": -45.0}, cfg)
    batched = fr.process_full_file_batched(e, x, sr, 15.0, 110.0, {"threshold_dB": -45.0}, cfg)
    assert len(one) > 50
    _same_calls(one, batched)
    assert {c["highpass_kHz"] for c in batched} >= {0.0, 20.0, 25.0, 30.0}
    # the measurement stage on top, both ways: identical rows
    cfg = dict(cfg, debugMode=True)
    one = fr.process_full_file(e, x, sr, 15.0, 110.0, {"threshold_dB": -45.0}, cfg, measure=True)
    batched = fr.process_full_file_batched(e, x, sr, 15.0, 110.0, {"threshold_dB": -45.0}, cfg, measure=True)
    assert len(one) > 50 and len(one) == len(batched)
    for a, b in zip(one, batched):
        assert {k: v for k, v in a.items() if k != "zonalNoiseMap"} == {k: v for k, v in b.items() if k != "zonalNoiseMap"}
    import spectrogram_amd.framing as frm
    small = frm.process_full_file_native(e, x[: sr // 2], sr, 15.0, 110.0, {"threshold_dB": -45.0}, cfg)      # results of an earlier run
    native = frm.process_full_file_native(e, x, sr, 15.0, 110.0, {"threshold_dB": -45.0}, cfg, threads=3)
    assert 0 < len(small) < len(native) == len(batched)
    assert all(a == {k: v for k, v in b.items() if k != "zonalNoiseMap"} for a, b in zip(native, batched))
    e.free()


def test_full_file_in_time_expansion_mode():
    """getTimeExpansionMode() on (fileState.js:158-166): every measured call is rescaled by 10 at the end of
    measureFrequencyParameters (batCallDetector.js:3890-3893), BEFORE processFullFile's <= 1 ms filter, Flow / Fhigh,
    the bandwidth classes and the (unscaled) region offset (js:802-825).  The three forms of the whole-file pass agree
    bit for bit, and against the run with the switch off: frequencies * 10, durations / 10, calls of 10 ms or less gone."""
    sr = 256000
    rng = np.random.default_rng(5)
    x = (0.005 * rng.standard_normal(3 * sr)).astype(np.float32)
    for i, k in enumerate(range(20000, 3 * sr - 20000, 40000)):
        n = [1200, 2600, 5200, 9000][i % 4]                      # 4.7 to 35 ms: some calls fall under 10 ms
        t = np.arange(n) / sr
        x[k:k + n] += (0.4 * np.hanning(n) * np.sin(2 * np.pi * (60000 * t - 2.5e4 / t[-1] * t * t / 2))).astype(np.float32)
    e = sa.SpectrogramEngine(1024, "hann")
    args = (e, x, sr, 10.0, 128.0, {"threshold_dB": -50.0})
    plain = fr.process_full_file_native(*args, {"debugMode": True})
    cfg = {"debugMode": True, "timeExpansion": True}
    native = fr.process_full_file_native(*args, cfg)
    batched = fr.process_full_file_batched(*args, cfg, measure=True)
    one = fr.process_full_file(*args, cfg, measure=True)
    strip = lambda c: {k: v for k, v in c.items() if k != "zonalNoiseMap"}
    assert [strip(c) for c in one] == [strip(c) for c in batched] == native
    key = lambda c: (c["roi"], c["startFrame"], c["endFrame"])
    by_key = {key(c): c for c in plain}
    survivors = [c for c in plain if c["duration_ms"] / 10 > 1.0]
    assert 0 < len(survivors) < len(plain), (len(survivors), len(plain))
    assert [key(c) for c in native] == [key(c) for c in survivors]
    for c in native:
        p = by_key[key(c)]
        for k in ("peakFreq_kHz", "highFreq_kHz", "lowFreq_kHz", "startFreq_kHz", "endFreq_kHz", "characteristicFreq_kHz",
                  "bandwidth_kHz", "kneeFreq_kHz", "heelFreq_kHz"):
            assert c[k] == (None if p[k] is None else p[k] * 10), k
        for k in ("duration_ms", "peakFreq_ms", "highFreq_ms", "endFreq_ms", "lowFreq_ms", "kneeFreq_ms"):
            assert c[k] == (None if p[k] is None else p[k] / 10), k
        assert c["Flow"] == c["lowFreq_kHz"] * 1000 and c["Fhigh"] == c["highFreq_kHz"]
        assert c["peakPower_dB"] == p["peakPower_dB"] and c["snr_dB"] == p["snr_dB"]
        assert abs((c["endTime_s"] - c["startTime_s"]) - (p["endTime_s"] - p["startTime_s"]) / 10) < 1e-9
        assert c["callType"] == ("CF" if c["bandwidth_kHz"] < 5 else "FM" if c["bandwidth_kHz"] > 20 else "CF-FM")
    e.free()


def _mixed_calls(sr, seconds, seed):
    """Pulses of many shapes over noise: steep and shallow sweeps, with and without a second harmonic, quiet and loud,
    short and long, a constant-frequency tone with a sweep at its end -- enough to walk the measurement through its
    branches (both threshold searches' stops, the safety re-scan, knee and heel, discards)."""
    rng = np.random.default_rng(seed)
    x = (0.004 * rng.standard_normal(int(seconds * sr))).astype(np.float32)
    k = int(0.02 * sr)
    i = 0
    while k < len(x) - int(0.06 * sr):
        n = int(sr * [0.0025, 0.004, 0.008, 0.015, 0.03][i % 5])
        t = np.arange(n) / sr
        f0 = [95000, 70000, 52000, 40000, 28000, 84000][i % 6]
        f1 = f0 * [0.35, 0.5, 0.75, 0.97, 1.0][(i // 2) % 5]
        u = t / t[-1]
        phase = 2 * np.pi * (f0 * t + (f1 - f0) * t[-1] * (u ** 1.6) / 1.6)
        amp = [0.5, 0.12, 0.03, 0.3][i % 4]
        pulse = amp * np.hanning(n) ** [1.0, 0.3][(i // 3) % 2] * np.sin(phase)
        if i % 3 == 1:
            pulse += 0.4 * amp * np.hanning(n) * np.sin(2 * phase)
        x[k:k + n] += pulse.astype(np.float32)
        k += n + int(sr * [0.012, 0.05, 0.031, 0.08][i % 4])
        i += 1
    return x


def test_measurement_on_the_gpu_equals_the_host_stage():
    """The calls' measurement runs on the GPU where their dB matrices lie (one wave per call, sono_measure_dev.hip); the
    engine option "host_measure" sends the matrices to the host threads instead.  Both are the same source
    (sono_measure_core.h): every field of every call, the discards and the reference's TypeError equal bit for bit."""
    import spectrogram_amd.framing as frm
    from spectrogram_amd import measure as ms
    sr = 384000
    e = sa.SpectrogramEngine(1024, "hann")
    seen = threw = 0
    for seed, seconds, band in ((3, 20.0, (10.0, 128.0)), (4, 8.0, (18.0, 100.0))):
        x = _mixed_calls(sr, seconds, seed)
        for cfg in ({"debugMode": True}, {"debugMode": True, "timeExpansion": True}, {}):
            got = []
            for host in (True, False):
                e.set_option("host_measure", host)
                try:
                    got.append(frm.process_full_file_table(e, x, sr, band[0], band[1], {"threshold_dB": -50.0}, cfg, file_id=2))
                except ms.ReferenceThrows:
                    got.append("throws")
            if isinstance(got[0], str) or isinstance(got[1], str):
                assert got[0] == got[1] == "throws", cfg
                threw += 1
                continue
            assert got[0].shape == got[1].shape and got[0].shape[0] > 20, got[0].shape
            assert np.array_equal(got[0].view(np.uint64), got[1].view(np.uint64)), cfg         # NaNs included
            seen += got[0].shape[0]
    assert seen > 200, (seen, threw)
    # a minute of the same over louder noise and a weak tone: more calls that end in the noise, where the searches'
    # stops and the harmonics' detours are decided
    x = _mixed_calls(sr, 60.0, 11)
    rng = np.random.default_rng(12)
    x += (0.01 * rng.standard_normal(len(x))).astype(np.float32)
    x += (0.002 * np.sin(2 * np.pi * 61000 * np.arange(len(x)) / sr)).astype(np.float32)
    got = []
    for host in (True, False):
        e.set_option("host_measure", host)
        got.append(frm.process_full_file_table(e, x, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True}, file_id=3))
    assert got[0].shape == got[1].shape and got[0].shape[0] > 400, got[0].shape
    assert np.array_equal(got[0].view(np.uint64), got[1].view(np.uint64))
    # the 16-bit entry point takes the same way
    q = np.clip(np.round(_mixed_calls(sr, 6.0, 9) * 32768), -32768, 32767).astype(np.int16)
    e.set_option("host_measure", True)
    host = frm.process_full_file_table(e, q, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True})
    e.set_option("host_measure", False)
    gpu = frm.process_full_file_table(e, q, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True})
    assert len(gpu) > 20 and np.array_equal(host.view(np.uint64), gpu.view(np.uint64))
    # a call longer than the GPU's work area is laid out for is measured on the host, in the same run
    t = np.arange(int(0.5 * sr)) / sr
    long_call = (0.003 * np.random.default_rng(1).standard_normal(int(0.8 * sr))).astype(np.float32)
    long_call[20000:20000 + len(t)] += (0.3 * np.hanning(len(t)) ** 0.2 * np.sin(2 * np.pi * 45000 * t)).astype(np.float32)
    rows = frm.process_full_file_table(e, long_call, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True})
    e.set_option("host_measure", True)
    assert np.array_equal(rows.view(np.uint64), frm.process_full_file_table(e, long_call, sr, 10.0, 128.0, {"threshold_dB": -50.0},
                                                                             {"debugMode": True}).view(np.uint64))
    e.free()


def test_upload_beside_the_analysis_changes_nothing(golden):
    """The recording crosses the link in pieces, and the regions of the pieces that have arrived are analysed, framed and
    measured while the next pieces are on their way.  With pieces of 64 fft sizes (engine option "small_upload_pieces")
    a few seconds of audio are cut into dozens of pieces -- runs and regions straddle the cuts, some pieces hold no
    region at all -- and every call equals the one-piece form's ("no_upload_overlap") bit for bit, region numbers
    included; so does the reference's TypeError.  The groups of regions (six at a time here) have their calls framed and
    measured on a thread (the measurement on a companion engine) beside the next group's analysis, or in line
    ("no_measure_overlap"): same calls, same error."""
    import spectrogram_amd.framing as frm
    from spectrogram_amd import measure as ms
    g, meta = golden
    sr = 384000
    e = sa.SpectrogramEngine(1024, "hann")
    recordings = [(_mixed_calls(sr, 7.0, 23), sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True}),
                  (_mixed_calls(sr, 5.0, 24), sr, 15.0, 110.0, {"threshold_dB": -45.0, "padding_ms": 12.0}, {"debugMode": True, "hopPercent": 6.25}),
                  (np.zeros(3 * sr, np.float32), sr, 10.0, 128.0, {"threshold_dB": -50.0}, {}),
                  (_mixed_calls(sr, 4.0, 25), sr, 10.0, 128.0, {"threshold_dB": -50.0}, {})]
    loud = (0.2 * np.sin(2 * np.pi * 40000 * np.arange(2 * sr) / sr)).astype(np.float32)          # one run from the first frame to the last
    recordings.append((loud, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True}))
    (c,) = cases(meta, "fullfile_measured")
    recordings.append((g[c["name"] + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"],
                       {"threshold_dB": c["scan_threshold_db"], "padding_ms": c["padding_ms"]},
                       {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"],
                        "minCallDuration_ms": c["min_call_duration_ms"], "debugMode": True}))
    rows_seen = 0
    for x, rate, flow, fhigh, opt, cfg in recordings:
        for samples in (x, np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)):
            got = []
            # one piece; pieces, a group's calls framed and measured beside the next group's analysis; pieces, one thing at a time
            for pieces, in_line in ((False, False), (True, False), (True, True)):
                e.set_option("no_upload_overlap", not pieces)
                e.set_option("small_upload_pieces", pieces)
                e.set_option("no_measure_overlap", in_line)
                try:
                    native = frm.process_full_file_native(e, samples, rate, flow, fhigh, opt, cfg)
                    got.append(native)
                except ms.ReferenceThrows as err:
                    got.append(str(err))
            assert got[0] == got[1] == got[2], (len(x), opt, cfg)
            rows_seen += len(got[0]) if isinstance(got[0], list) else 0
    assert rows_seen > 200
    # with the calls' spectrograms kept (the measurement then runs on host threads, and the tail hands the rows on)
    x, rate, flow, fhigh, opt, cfg = recordings[0]
    kept = []
    for pieces, in_line in ((False, False), (True, False), (True, True)):
        e.set_option("no_upload_overlap", not pieces)
        e.set_option("small_upload_pieces", pieces)
        e.set_option("no_measure_overlap", in_line)
        kept.append(frm.process_full_file_native(e, x, rate, flow, fhigh, opt, cfg, keep_spectrograms=True))
    assert len(kept[0]) > 20 and len(kept[0]) == len(kept[1]) == len(kept[2])
    for a, b, c in zip(*kept):
        assert a.keys() == b.keys() == c.keys()
        for key, va in a.items():
            for vo in (b[key], c[key]):
                if isinstance(va, (list, tuple, np.ndarray)):
                    assert len(va) == len(vo) and all(np.array_equal(np.asarray(p), np.asarray(q), equal_nan=True) for p, q in zip(va, vo)), key
                else:
                    assert va == vo or (va != va and vo != vo), key
    # one channel of an interleaved 16-bit recording, piece by piece
    x = _mixed_calls(sr, 3.0, 26)
    q = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
    inter = np.stack([np.zeros_like(q), q, (q // 3).astype(np.int16)], axis=1)                    # frames x 3 channels
    got = []
    for pieces in (False, True):
        e.set_option("no_upload_overlap", not pieces)
        e.set_option("small_upload_pieces", pieces)
        out, count = frm._full_file_call_array(e, inter, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True}, 0, False, pcm_channel=1)
        got.append(bytes(out)[: count * C.sizeof(sa.FullFileCall)])
    out, count = frm._full_file_call_array(e, q, sr, 10.0, 128.0, {"threshold_dB": -50.0}, {"debugMode": True}, 0, False)
    mono = bytes(out)[: count * C.sizeof(sa.FullFileCall)]
    assert count > 10 and got[0] == got[1] == mono
    e.set_option("no_upload_overlap", False)
    e.set_option("small_upload_pieces", False)
    e.set_option("no_measure_overlap", False)
    e.free()


def test_a_sample_rate_too_low_for_the_oscillogram_window_is_refused_on_every_way():
    """refineEndUsingOscillogram's RMS window is 0.1 ms with a hop of half of it (batCallDetector.js:1660-1664): below
    20 kHz the hop has no sample and the reference's loop would never advance.  The library refuses such a recording
    with SONO_EINVAL wherever the refinement is reached: in the whole-file detection, whose windows are summed on the
    GPU -- piece by piece, with a group's calls framed on a thread of its own, and in one piece --, and in the batched
    form that frames on the host."""
    import spectrogram_amd.framing as frm
    sr = 16000
    rng = np.random.default_rng(5)
    x = (0.001 * rng.standard_normal(4 * sr)).astype(np.float32)
    for k in range(8):                                                            # loud bursts of 30 ms at 4 kHz
        at = int((0.2 + 0.45 * k) * sr)
        n = int(0.03 * sr)
        x[at:at + n] += (0.5 * np.hanning(n) * np.sin(2 * np.pi * 4000 * np.arange(n) / sr)).astype(np.float32)
    e = sa.SpectrogramEngine(256, "hann")
    opt, cfg = {"threshold_dB": -40.0}, {"debugMode": True}
    for pieces, in_line in ((False, False), (True, False), (True, True)):
        e.set_option("no_upload_overlap", not pieces)
        e.set_option("small_upload_pieces", pieces)
        e.set_option("no_measure_overlap", in_line)
        with pytest.raises(sa.SonoError) as err:
            frm.process_full_file_native(e, x, sr, 1.0, 7.0, opt, cfg)
        assert err.value.status == sa.SONO_EINVAL and "20 kHz" in str(err.value), err.value
    with pytest.raises(sa.SonoError) as err:
        frm.process_full_file_batched(e, x, sr, 1.0, 7.0, opt, cfg)
    assert err.value.status == sa.SONO_EINVAL and "20 kHz" in str(err.value), err.value
    # the engine is as good as new afterwards
    e.set_option("no_upload_overlap", False)
    e.set_option("small_upload_pieces", False)
    e.set_option("no_measure_overlap", False)
    y = _mixed_calls(384000, 2.0, 31)
    e2 = sa.SpectrogramEngine(1024, "hann")
    assert len(frm.process_full_file_native(e2, y, 384000, 10.0, 128.0, {"threshold_dB": -50.0}, cfg)) > 5
    e.free()
    e2.free()


def test_rms_based_snr(golden):
    g, meta = golden
    e = sa.SpectrogramEngine(1024, "hann")
    for c in cases(meta, "snr"):
        n = c["name"]
        F, nb = (int(v) for v in g[n + "/shape"])
        db, fb = g[n + "/db"].reshape(F, nb), g[n + "/freq"]
        noise = None
        if n + "/noise_db" in g.files:
            noise = {"powerMatrix": g[n + "/noise_db"].reshape(-1, nb), "freqBins": fb}
        want = g[n + "/snr"].reshape(-1, 9)
        k = 0
        for q in c["queries"]:
            for use in (False, True):
                if use and noise is None:
                    continue
                r = fr.calculate_rms_based_snr(e, {"lowFreq_kHz": q[0], "highFreq_kHz": q[1]}, db, fb, q[2], q[3],
                                               c["flow_khz"], c["fhigh_khz"], noise if use else None)
                w = want[k]
                k += 1
                if np.isnan(w[0]):
                    assert r["snr_dB"] is None, (n, q, r)
                elif np.isinf(w[0]):
                    assert r["snr_dB"] == float("inf")
                else:
                    assert abs(r["snr_dB"] - w[0]) <= 1e-10 * max(1.0, abs(w[0])), (n, q, r, w)
                    assert abs(r["signalPowerMean_dB"] - w[1]) <= 1e-10 and abs(r["noisePowerMean_dB"] - w[2]) <= 1e-10
                    assert (r["signalCount"], r["noiseCount"]) == (int(w[3]), int(w[4]))
                if not np.isnan(w[5]):
                    assert (r["signalThreshold"], r["signalMax"], r["signalMin"]) == (w[5], w[6], w[7])
        assert k == len(want)
    e.free()


def _quantise(x):
    return np.clip(np.round(np.asarray(x, np.float64) * 32767.0), -32768, 32767).astype(np.int16)


def test_whole_file_detection_from_16_bit_samples(golden):
    """sono_process_full_file_pcm16 takes the recording as a WAV file holds it (int16 sample frames, one channel of
    them analysed; the samples mean value / 32768).  Its calls must equal, field for field, those of the f32 entry point
    on the same quantised samples converted to float -- on the recorded whole-file cases quantised to 16 bits, on a
    recording with regions of every class of the automatic high-pass (filtered regions are read back from the GPU,
    unfiltered ones are converted from the integers on the host), mono and as channel 1 of a three-channel file."""
    import spectrogram_amd.framing as frm
    g, meta = golden
    runs = []
    for c in cases(meta, "fullfile") + cases(meta, "fullfile_measured"):
        n = c["name"]
        cfg = {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"], "minCallDuration_ms": c["min_call_duration_ms"],
               "debugMode": True}
        runs.append((c["fft"], c["window"], g[n + "/x"], c["sr"], c["flow_khz"], c["fhigh_khz"],
                     {"threshold_dB": c["scan_threshold_db"], "padding_ms": c["padding_ms"]}, cfg))
    sr = 384000
    rng = np.random.default_rng(12)
    x = (0.01 * rng.standard_normal(3 * sr)).astype(np.float32)
    t = np.arange(1536) / sr
    for k in range(0, 3 * sr - 4000, 19200):
        f0 = [80000, 38000, 33000, 25000][(k // 19200) % 4]
        x[k + 500:k + 2036] += (0.4 * np.hanning(1536) * np.sin(2 * np.pi * (f0 * t - 1e5 * t * t))).astype(np.float32)
    runs.append((1024, "hann", x, sr, 15.0, 110.0, {"threshold_dB": -45.0}, {"hopPercent": 6.25, "debugMode": True}))
    total = 0
    for fft, window, x, sr, flow, fhigh, opt, cfg in runs:
        q = _quantise(x)
        xf = q.astype(np.float32) / np.float32(32768.0)
        e = sa.SpectrogramEngine(fft, window)
        want = frm.process_full_file_native(e, xf, sr, flow, fhigh, opt, cfg)
        got = frm.process_full_file_native(e, q, sr, flow, fhigh, opt, cfg)
        assert got == want
        three = np.stack([_quantise(0.3 * x[::-1]), q, np.zeros_like(q)], axis=1)
        assert frm._full_file_call_array(e, three, sr, flow, fhigh, opt, cfg, 0, False, pcm_channel=1)[1] == len(want)
        rows = frm.process_full_file_table(e, q, sr, flow, fhigh, opt, cfg, file_id=3, channel=1)
        assert np.array_equal(rows, frm.process_full_file_table(e, xf, sr, flow, fhigh, opt, cfg, file_id=3, channel=1), equal_nan=True)
        total += len(want)
        e.free()
    assert total > 30
    assert {c["highpass_kHz"] for c in want} >= {0.0, 20.0, 25.0, 30.0}
