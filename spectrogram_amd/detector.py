"""Detector front end on the GPU: the array passes of the reference's
BatCallDetector (modules/batCallDetector.js) that sit directly on the STFT
output, under the reference's method names (snake_case), argument order and
return shapes.  Everything numeric runs in libsonostft; nothing is computed on
the CPU here except the trivial axis formulas.
"""
import ctypes as C

import numpy as np

from . import FrameFeatures as _FrameFeatures, Segment as _Segment, _check, _f32, _p, lib

__all__ = ["fast_scan_segments", "merge_and_pad_segments", "generate_spectrogram", "detect_call_segments",
           "process_full_file_front_end",
           "frame_features", "DEFAULT_DETECTION_CONFIG"]

# the two members of DEFAULT_DETECTION_CONFIG (batCallDetector.js:3-45) these passes read
DEFAULT_DETECTION_CONFIG = {"callThreshold_dB": -24, "hopPercent": 3.125}


def fast_scan_segments(engine, audio_data, sample_rate, flow_khz, fhigh_khz, threshold_db):
    """fastScanSegmentsWasm (batCallDetector.js:878-949) -> [{'start': sample, 'end': sample}, ...]."""
    L = lib()
    x = _f32(audio_data)
    n_fft = engine.get_fft_size()
    cap = x.size // max(n_fft // 2, 1) + 2
    out = (_Segment * cap)()
    got = C.c_size_t(0)
    _check(L.sono_fast_scan_segments(engine._handle(), _p(x), x.size, float(sample_rate), float(flow_khz),
                                     float(fhigh_khz), float(threshold_db), out, cap, C.byref(got)))
    return [{"start": int(out[i].start), "end": int(out[i].end)} for i in range(got.value)]


def merge_and_pad_segments(segments, total_samples, sample_rate, padding_ms):
    """mergeAndPadSegments (batCallDetector.js:1028-1056) -> [{'startSample', 'endSample'}, ...]."""
    L = lib()
    n = len(segments)
    if n == 0:
        return []
    segs = (_Segment * n)(*[_Segment(int(s["start"]), int(s["end"])) for s in segments])
    out = (_Segment * n)()
    got = C.c_size_t(0)
    _check(L.sono_merge_and_pad_segments(segs, n, int(total_samples), float(sample_rate), float(padding_ms), out, n,
                                         C.byref(got)))
    return [{"startSample": int(out[i].start), "endSample": int(out[i].end)} for i in range(got.value)]


def generate_spectrogram(engine, audio_data, sample_rate, flow_khz, fhigh_khz, hop_percent=None):
    """generateSpectrogramWasm (batCallDetector.js:1296-1377) ->
    {'powerMatrix': frames x bins f32 dB, 'timeFrames', 'freqBins', 'freqResolution'} or None."""
    L = lib()
    if hop_percent is None:
        hop_percent = DEFAULT_DETECTION_CONFIG["hopPercent"]
    x = _f32(audio_data)
    n_fft = engine.get_fft_size()
    hop = int(np.floor(n_fft * (hop_percent / 100)))
    if hop < 1 or n_fft > x.size:
        return None
    frames_cap = (x.size - n_fft) // hop + 1
    out = np.empty(frames_cap * (n_fft // 2), dtype=np.float32)
    frames, min_bin, nb = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0)
    _check(L.sono_generate_spectrogram(engine._handle(), _p(x), x.size, float(sample_rate), float(flow_khz),
                                       float(fhigh_khz), float(hop_percent), _p(out), out.size, C.byref(frames),
                                       C.byref(min_bin), C.byref(nb)))
    if frames.value == 0:
        return None
    F, B = frames.value, nb.value
    freq_res = sample_rate / n_fft
    return {
        "powerMatrix": out[: F * B].reshape(F, B),
        "timeFrames": (np.arange(F) * hop + n_fft / 2) / sample_rate,                 # js:1350-1351
        "freqBins": ((min_bin.value + np.arange(B)) * freq_res).astype(np.float32),   # js:1342-1344
        "freqResolution": freq_res,
    }


def detect_call_segments(engine, power_matrix, call_threshold_db=None):
    """detectCallSegments (batCallDetector.js:1540-1598) -> [{'startFrame', 'endFrame'}, ...]."""
    L = lib()
    if call_threshold_db is None:
        call_threshold_db = DEFAULT_DETECTION_CONFIG["callThreshold_dB"]
    p = np.ascontiguousarray(power_matrix, dtype=np.float32)
    F, B = (p.shape if p.ndim == 2 else (0, 0))
    cap = F // 2 + 1
    out = (_Segment * cap)()
    got = C.c_size_t(0)
    _check(L.sono_detect_call_segments(engine._handle(), _p(p), F, B, float(call_threshold_db), out, cap, C.byref(got)))
    return [{"startFrame": int(out[i].start), "endFrame": int(out[i].end)} for i in range(got.value)]


def frame_features(engine, power_matrix, peak_b0, peak_b1, low_max_bin, low_threshold_db, high_min_bin,
                   high_threshold_db):
    """Per-frame scans of measureFrequencyParameters (batCallDetector.js:3520-3527, 3612-3630, 2509-2522,
    2002-2038) -> (peak_bin i32, peak_db f32, low_bin i32, high_bin i32), -1 / -inf where nothing qualifies."""
    L = lib()
    p = np.ascontiguousarray(power_matrix, dtype=np.float32)
    F, B = p.shape
    out = (_FrameFeatures * max(F, 1))()
    _check(L.sono_frame_features_db(engine._handle(), _p(p), F, B, int(peak_b0), int(peak_b1), int(low_max_bin),
                                    float(low_threshold_db), int(high_min_bin), float(high_threshold_db), out, F))
    a = np.frombuffer(out, dtype=np.dtype([("peak_bin", "<i4"), ("peak_db", "<f4"), ("low_bin", "<i4"),
                                           ("high_bin", "<i4")]), count=F)
    return a["peak_bin"].copy(), a["peak_db"].copy(), a["low_bin"].copy(), a["high_bin"].copy()


def process_full_file_front_end(engine, full_audio, sample_rate, flow_khz, fhigh_khz, threshold_db=-60, padding_ms=5,
                                hop_percent=None, call_threshold_db=None, min_call_duration_ms=0.0):
    """The array stages of processFullFile (batCallDetector.js:584-735) in their order: fast scan ->
    merge & pad -> per ROI: analysis spectrogram -> call segments -> peak of every segment.  The serial
    measurement stage that follows in the reference (auto high-pass, measureFrequencyParameters) is not
    part of this library.  -> list of candidate calls:
    {'roi', 'startSample', 'startFrame', 'endFrame', 'startTime_s', 'endTime_s', 'duration_ms',
     'peakFrame', 'peakTime_s', 'peakFreq_kHz', 'peakPower_dB'} with times measured from the start of the file."""
    x = _f32(full_audio)
    raw = fast_scan_segments(engine, x, sample_rate, flow_khz, fhigh_khz, threshold_db)            # js:594
    if not raw:
        return []
    merged = merge_and_pad_segments(raw, x.size, sample_rate, padding_ms)                          # js:601
    calls = []
    for roi, seg in enumerate(merged):
        audio = x[seg["startSample"]:seg["endSample"]]                                             # js:612
        spec = generate_spectrogram(engine, audio, sample_rate, flow_khz, fhigh_khz, hop_percent)  # js:617
        if spec is None:
            continue
        power, t, fb = spec["powerMatrix"], spec["timeFrames"], spec["freqBins"]
        segments = detect_call_segments(engine, power, call_threshold_db)                          # js:677
        if not segments:
            continue
        nb = power.shape[1]
        peak_bin, peak_db, _, _ = frame_features(engine, power, 0, nb - 1, 0, np.inf, nb, np.inf)
        t0 = seg["startSample"] / sample_rate
        for s in segments:
            f0, f1 = s["startFrame"], s["endFrame"]
            if t[f1] - t[f0] < min_call_duration_ms / 1000:                                         # js:688-690
                continue
            pf = f0 + int(np.argmax(peak_db[f0:f1 + 1]))           # first frame holding the segment's largest bin (js:696-705)
            calls.append({
                "roi": roi, "startSample": seg["startSample"], "startFrame": f0, "endFrame": f1,
                "startTime_s": t0 + t[f0], "endTime_s": t0 + t[f1], "duration_ms": (t[f1] - t[f0]) * 1000,
                "peakFrame": pf, "peakTime_s": t0 + t[pf], "peakFreq_kHz": float(fb[peak_bin[pf]]) / 1000,
                "peakPower_dB": float(peak_db[pf]),
            })
    return calls
