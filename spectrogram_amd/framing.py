"""Call framing: what the reference's BatCallDetector.detectCalls
(modules/batCallDetector.js:1070-1289) does between detectCallSegments and the
measurement stage, under the reference's method names (snake_case).  The matrix
passes (zonal histograms, the noise-floor order statistic, call segments) run on
the GPU; the per-call serial steps (oscillogram refinement, Butterworth
high-pass) are host C++ in libsonostft.  Nothing numeric is computed here.
"""
import ctypes as C
import math

import numpy as np

from . import SnrResult as _SnrResult
from . import (CallFrame as _CallFrame, RoiCall as _RoiCall, RoiTable as _RoiTable, Segment as _Segment, SONO_MAX_ZONES, _check,
               _f32, _p, lib)
from . import detector as _det

__all__ = ["calculate_zonal_noise_floors", "db_percentile", "refine_end_using_oscillogram", "frame_calls",
           "scan_spectrum_for_frequency", "calculate_auto_highpass_filter_freq", "apply_highpass_filter",
           "savitzky_golay", "detect_calls_framing", "detect_calls", "process_full_file", "process_full_file_batched", "process_full_file_native",
           "calculate_rms_based_snr",
           "DEFAULT_FRAMING_CONFIG"]

# the members of DEFAULT_DETECTION_CONFIG (batCallDetector.js:3-45) this stage reads
DEFAULT_FRAMING_CONFIG = {"minCallDuration_ms": 1, "callThreshold_dB": -24, "hopPercent": 3.125}


def calculate_zonal_noise_floors(engine, spectrogram, freq_bins, start_frame, end_frame):
    """calculateZonalNoiseFloors (batCallDetector.js:1818-1924) -> {zone start in kHz (int): floor dB}."""
    p = np.ascontiguousarray(spectrogram, dtype=np.float32)
    fb = _f32(freq_bins)
    F, B = (p.shape if p.ndim == 2 else (0, fb.size))
    floors = (C.c_double * 256)()
    zones = C.c_size_t(0)
    if end_frame < start_frame:                      # empty range: every zone gets the default (the loop never runs)
        start_frame, end_frame = 1, 0
    _check(lib().sono_zonal_noise_floors(engine._handle(), _p(p) if p.size else None, F, B, _p(fb), max(0, int(start_frame)),
                                         max(0, int(end_frame)), floors, 256, C.byref(zones)))
    return {10 * z: floors[z] for z in range(zones.value)}


def db_percentile(engine, spectrogram, fraction=0.25):
    """The order statistic detectCalls uses as noise floor (batCallDetector.js:1223-1233)."""
    p = np.ascontiguousarray(spectrogram, dtype=np.float32).ravel()
    v = C.c_float(0)
    _check(lib().sono_db_percentile(engine._handle(), _p(p), p.size, float(fraction), C.byref(v)))
    return float(v.value)


def refine_end_using_oscillogram(audio_data, sample_rate, start_sample, end_sample):
    """refineEndUsingOscillogram (batCallDetector.js:1646-1803) -> end sample."""
    x = _f32(audio_data)
    r = C.c_int64(0)
    _check(lib().sono_refine_end_oscillogram(_p(x), x.size, float(sample_rate), int(start_sample), int(end_sample), C.byref(r)))
    return int(r.value)


def frame_calls(segments, time_frames, audio_data, sample_rate, fft_size, min_call_duration_ms=None, padding_ms=3.0,
                full_file_rules=False):
    """The per-segment part of detectCalls (batCallDetector.js:1128-1208) or, with full_file_rules and
    padding_ms=5, of processFullFile (batCallDetector.js:739-810) ->
    [{'startFrame', 'endFrame', 'startTime_s', 'endTime_s', 'duration_ms'}, ...] in the order of `segments`."""
    if min_call_duration_ms is None:
        min_call_duration_ms = DEFAULT_FRAMING_CONFIG["minCallDuration_ms"]
    n = len(segments)
    if n == 0:
        return []
    x = _f32(audio_data)
    tf = np.ascontiguousarray(time_frames, dtype=np.float64)
    segs = (_Segment * n)(*[_Segment(int(s["startFrame"]), int(s["endFrame"])) for s in segments])
    out = (_CallFrame * n)()
    got = C.c_size_t(0)
    _check(lib().sono_frame_calls(segs, n, tf.ctypes.data_as(C.POINTER(C.c_double)), tf.size, _p(x), x.size, float(sample_rate),
                                  int(fft_size), float(padding_ms), float(min_call_duration_ms), 1 if full_file_rules else 0,
                                  out, n, C.byref(got)))
    return [{"startFrame": int(out[i].start_frame), "endFrame": int(out[i].end_frame), "startTime_s": out[i].start_time_s,
             "endTime_s": out[i].end_time_s, "duration_ms": out[i].duration_ms} for i in range(got.value)]


def scan_spectrum_for_frequency(spectrum_data, freq_bins, threshold_db, direction):
    """_scanSpectrumForFrequency (batCallDetector.js:2936-2987) -> {'freq_Hz', 'binIdx'} or None."""
    row, fb = _f32(spectrum_data), _f32(freq_bins)
    f, b = C.c_double(0), C.c_int64(0)
    _check(lib().sono_scan_spectrum_for_frequency(_p(row), _p(fb), row.size, float(threshold_db),
                                                  1 if direction == "HighToLow" else 0, C.byref(f), C.byref(b)))
    return None if b.value < 0 else {"freq_Hz": f.value, "binIdx": int(b.value)}


def calculate_auto_highpass_filter_freq(peak_freq_khz):
    """calculateAutoHighpassFilterFreq (batCallDetector.js:4028-4035) -> kHz."""
    return lib().sono_auto_highpass_khz(float(peak_freq_khz))


def apply_highpass_filter(audio_data, filter_freq_hz, sample_rate, order=4):
    """applyHighpassFilter (batCallDetector.js:4045-4140) -> float32 array."""
    x = _f32(audio_data)
    out = np.empty_like(x)
    _check(lib().sono_highpass_filter(_p(x), x.size, float(filter_freq_hz), float(sample_rate), float(order), _p(out)))
    return out


def savitzky_golay(data):
    """savitzkyGolay (batCallDetector.js:1610-1640) -> list of floats."""
    d = np.ascontiguousarray(data, dtype=np.float64)
    out = np.empty_like(d)
    dp = C.POINTER(C.c_double)
    _check(lib().sono_savitzky_golay(d.ctypes.data_as(dp), d.size, out.ctypes.data_as(dp)))
    return out.tolist()


def detect_calls_framing(engine, audio_data, sample_rate, flow_khz, fhigh_khz, config=None, noise_spectrogram=None):
    """detectCalls (batCallDetector.js:1070-1289) up to the measurement stage ->
    (calls, zonal_noise_map, noise_floor_dB, spectrogram): calls as frame_calls returns them, each with
    'noiseFloor_dB'; zonal_noise_map = the map the reference hands to measureFrequencyParameters;
    noise_floor_dB = max(25th percentile, -80); spectrogram = generate_spectrogram's dict (None if there is none)."""
    cfg = dict(DEFAULT_FRAMING_CONFIG, **(config or {}))
    x = _f32(audio_data)
    sp = _det.generate_spectrogram(engine, x, sample_rate, flow_khz, fhigh_khz, cfg["hopPercent"])
    if sp is None:
        return [], {}, None, None
    p, tf, fb = sp["powerMatrix"], sp["timeFrames"], sp["freqBins"]
    ns = noise_spectrogram
    if ns is not None and len(ns["powerMatrix"]) > 0:                                             # js:1086
        total, first = len(ns["powerMatrix"]), 0
        if len(ns["timeFrames"]) > 1:
            per = float(ns["timeFrames"][1] - ns["timeFrames"][0])
            first = max(0, total - math.ceil(0.005 / per))                                        # js:1093-1098
        zonal = calculate_zonal_noise_floors(engine, ns["powerMatrix"], ns["freqBins"], first, total - 1)
    else:
        zonal = calculate_zonal_noise_floors(engine, p, fb, 0, min(5, len(p) - 1))               # js:1115-1120
    segments = _det.detect_call_segments(engine, p, cfg["callThreshold_dB"])                       # js:1124
    if not segments:
        return [], zonal, None, sp
    calls = frame_calls(segments, tf, x, sample_rate, engine.get_fft_size(), cfg["minCallDuration_ms"])
    floor = max(db_percentile(engine, p, 0.25), -80.0)                                             # js:1232-1235
    for c in calls:
        c["noiseFloor_dB"] = floor
    return calls, zonal, floor, sp


def detect_calls(engine, audio_data, sample_rate, flow_khz, fhigh_khz, options=None, config=None):
    """detectCalls (batCallDetector.js:1070-1289), whole: framing, the measurement stage on every call, classification,
    and the noise-floor / SNR loop.  options: {'skipSNR': False, 'noiseSpectrogram': None} as in the reference;
    config['debugMode'] is the detector's debug switch (outside it the reference raises on some calls:
    measure.ReferenceThrows), config['timeExpansion'] the global getTimeExpansionMode() (fileState.js:158-166).  -> list of call dicts with the reference's BatCall fields.  Unlike processFullFile the
    reference does not look at isDiscarded here: such calls stay in the list, flagged, with null frequencies."""
    from . import measure as _ms
    opt = dict({"skipSNR": False, "noiseSpectrogram": None}, **(options or {}))
    cfg = dict(DEFAULT_FRAMING_CONFIG, **(config or {}))
    framed, zonal, floor, sp = detect_calls_framing(engine, audio_data, sample_rate, flow_khz, fhigh_khz, cfg, opt["noiseSpectrogram"])
    if not framed:
        return []
    p, tf, fb = sp["powerMatrix"], sp["timeFrames"], sp["freqBins"]
    calls = []
    for k in framed:                                                                                   # js:1139-1218
        f0, f1 = k["startFrame"], k["endFrame"]
        call = {"spectrogram": p[f0:f1 + 1], "timeFrames": tf[f0:f1 + 2], "startTime_s": k["startTime_s"],
                "endTime_s": k["endTime_s"], "duration_ms": k["duration_ms"]}
        _ms.measure_frequency_parameters(call, flow_khz, fhigh_khz, fb, sp["freqResolution"], zonal, cfg.get("debugMode", False),
                                         cfg.get("timeExpansion", False))
        call["Flow"] = (call["lowFreq_kHz"] or 0.0) * 1000                                             # null * 1000 is 0 in JavaScript
        call["Fhigh"] = call["highFreq_kHz"]
        call["callType"] = _ms.classify_call_type(call)
        call.update(startFrame=f0, endFrame=f1)
        calls.append(call)
    kept = []
    for call in calls:                                                                                 # js:1238-1286
        if call["peakPower_dB"] is None:
            continue
        call["noiseFloor_dB"] = floor
        spectral = call["peakPower_dB"] - floor
        if opt["skipSNR"]:
            call.update(snr_dB=spectral, snrMechanism="Skipped (Filtered Pass)", quality=_ms.get_quality_rating(spectral))
            kept.append(call)
            continue
        box = {"lowFreq_kHz": call["lowFreq_kHz"] or 0.0, "highFreq_kHz": call["highFreq_kHz"] or 0.0}
        r = calculate_rms_based_snr(engine, box, call["spectrogram"], fb, 0, call["endFrameIdx_forLowFreq"] or 0, flow_khz, fhigh_khz,
                                    opt["noiseSpectrogram"])
        if r["snr_dB"] is not None and math.isfinite(r["snr_dB"]):
            call.update(snr_dB=r["snr_dB"], snrMechanism=r["mechanism"])
        else:
            call.update(snr_dB=spectral, snrMechanism="RMS-based (2025) - Calculation failed fallback")
        call["quality"] = _ms.get_quality_rating(call["snr_dB"])
        if spectral < -20:                                                                             # js:1280-1283
            continue
        kept.append(call)
    return kept


def _measure_framed(power, tf, fb, framed, flow_khz, fhigh_khz, freq_res, zonal, offset, debug_mode, time_expansion=False,
                    keep_spectrogram=False):
    """One framed call of a region through measureFrequencyParameters and processFullFile's follow-up
    (batCallDetector.js:786-827) -> the finished call dict, or None where the reference drops the call."""
    from . import measure as _ms
    f0, f1 = framed["startFrame"], framed["endFrame"]
    call = {"spectrogram": power[f0:f1 + 1], "timeFrames": tf[f0:f1 + 2], "startTime_s": framed["startTime_s"],
            "endTime_s": framed["endTime_s"], "duration_ms": framed["duration_ms"]}
    _ms.measure_frequency_parameters(call, flow_khz, fhigh_khz, fb, freq_res, zonal, debug_mode, time_expansion)
    call = _ms.finish_full_file_call(call, offset)
    if call is not None:
        if keep_spectrogram:
            call["freqBins"] = fb                                                                   # js:787
        else:
            del call["spectrogram"], call["timeFrames"]
        call.update(startFrame=f0, endFrame=f1, zonalNoiseMap=zonal)
    return call


def process_full_file(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options=None, config=None, measure=False,
                      keep_spectrograms=False):
    """processFullFile (batCallDetector.js:584-840).  measure=True runs the measurement stage on every framed call and
    returns the reference's BatCall fields (measure.CALL_FIELDS, 'Flow', 'Fhigh', 'callType', 'snr_dB', 'quality', frame
    numbers; config['debugMode'] is the detector's debug switch, outside which the reference -- and this -- raises on
    some calls: measure.ReferenceThrows; config['timeExpansion'] is the global getTimeExpansionMode(), under which every
    measured call is rescaled by 10 before the <= 1 ms filter, js:3890-3893; keep_spectrograms leaves 'spectrogram',
    'timeFrames' and 'freqBins' on the measured calls as the reference does, js:785-787).  measure=False stops before it: fast scan -> merge & pad ->
    per region: analysis spectrogram, automatic high-pass, call segments, strongest-first 30 ms echo filter,
    5 ms padding + oscillogram refinement -> calls in the reference's order (per region, strongest first):
    {'startTime_s', 'endTime_s', 'duration_ms' (file time), 'peakPower_dB', 'peakTime_s', 'peakFreq_kHz' and 'peakFreq_ms'
    (the first step of measureFrequencyParameters, js:3009-3045), 'snr_dB'
    (peak + 80, js:818), 'roi', 'startFrame', 'endFrame', 'highpass_kHz', 'zonalNoiseMap'}.
    options: {'threshold_dB': -60, 'padding_ms': 5} as in the reference."""
    opt = dict({"threshold_dB": -60, "padding_ms": 5}, **(options or {}))
    cfg = dict(DEFAULT_FRAMING_CONFIG, **(config or {}))
    x = _f32(full_audio_data)
    raw = _det.fast_scan_segments(engine, x, sample_rate, flow_khz, fhigh_khz, opt["threshold_dB"])          # js:594
    if not raw:
        return []
    merged = _det.merge_and_pad_segments(raw, x.size, sample_rate, opt["padding_ms"])                        # js:601
    calls = []
    for roi, seg in enumerate(merged):
        audio = x[seg["startSample"]:seg["endSample"]]                                                      # js:611
        spec = _det.generate_spectrogram(engine, audio, sample_rate, flow_khz, fhigh_khz, cfg["hopPercent"])
        if spec is None:
            continue
        power, tf, fb = spec["powerMatrix"], spec["timeFrames"], spec["freqBins"]
        nb = power.shape[1]
        zonal = calculate_zonal_noise_floors(engine, power, fb, 0, min(5, len(power) - 1))                  # js:622-627
        peak_bin, peak_db, _, _ = _det.frame_features(engine, power, 0, nb - 1, 0, np.inf, nb, np.inf)
        pf = int(np.argmax(peak_db))                                       # first frame holding the region's maximum, js:635-644
        cutoff = calculate_auto_highpass_filter_freq(float(fb[peak_bin[pf]]) / 1000)                         # js:646-647
        if cutoff > 0:                                                                                      # js:651-671
            audio = apply_highpass_filter(audio, cutoff * 1000, sample_rate)
            again = _det.generate_spectrogram(engine, audio, sample_rate, flow_khz, fhigh_khz, cfg["hopPercent"])
            if again is not None:
                power, tf, fb = again["powerMatrix"], again["timeFrames"], again["freqBins"]
                peak_bin, peak_db, _, _ = _det.frame_features(engine, power, 0, nb - 1, 0, np.inf, nb, np.inf)
        segments = _det.detect_call_segments(engine, power, cfg["callThreshold_dB"])                         # js:677
        candidates = []
        for s in segments:
            f0, f1 = s["startFrame"], s["endFrame"]
            if tf[f1] - tf[f0] < cfg["minCallDuration_ms"] / 1000:                                           # js:688-689
                continue
            k = f0 + int(np.argmax(peak_db[f0:f1 + 1]))                                                      # js:695-703
            candidates.append({"seg": s, "peakPower": float(peak_db[k]), "peakTime": float(tf[k]), "peakFrame": k})
        candidates.sort(key=lambda c: -c["peakPower"])                                                      # stable, js:714
        kept = []
        for c in candidates:                                                                                # js:720-734
            if all(abs(c["peakTime"] - k["peakTime"]) >= 0.030 for k in kept):
                kept.append(c)
        offset = seg["startSample"] / sample_rate                                                           # js:737
        for c in kept:
            framed = frame_calls([c["seg"]], tf, audio, sample_rate, engine.get_fft_size(), cfg["minCallDuration_ms"],
                                 padding_ms=5.0, full_file_rules=True)                                      # js:743-810
            for k in framed:
                if measure:
                    done = _measure_framed(power, tf, fb, k, flow_khz, fhigh_khz, spec["freqResolution"], zonal, offset,
                                           cfg.get("debugMode", False), cfg.get("timeExpansion", False), keep_spectrograms)
                    if done is not None:
                        calls.append(dict(done, roi=roi, highpass_kHz=cutoff))
                    continue
                f0, f1 = k["startFrame"], k["endFrame"]
                kk = f0 + int(np.argmax(peak_db[f0:f1 + 1]))
                khz = C.c_double(0)
                _check(lib().sono_peak_frequency_khz(_p(power[kk]), _p(fb), nb, int(peak_bin[kk]), C.byref(khz)))       # js:3021-3036
                bw = C.c_double(0)
                _check(lib().sono_instant_bandwidth_khz(_p(power[kk]), nb, float(peak_db[kk]), sample_rate / engine.get_fft_size(),
                                                        C.byref(bw)))                                           # js:3059-3077
                if bw.value > 20.0:                          # a vertical click: the reference drops the call (js:3080-3087, 802)
                    continue
                calls.append({
                    "roi": roi, "startFrame": f0, "endFrame": f1, "startTime_s": k["startTime_s"] + offset,
                    "endTime_s": k["endTime_s"] + offset, "duration_ms": k["duration_ms"],
                    "peakPower_dB": float(peak_db[kk]), "peakTime_s": float(tf[kk]) + offset,
                    "peakFreq_kHz": khz.value, "peakFreq_ms": (float(tf[kk]) - float(tf[f0])) * 1000,           # js:3036-3043
                    "instBandwidth_kHz": bw.value,
                    "snr_dB": float(peak_db[kk]) + 80.0,                                                       # js:818
                    "highpass_kHz": cutoff, "zonalNoiseMap": zonal,
                })
    return calls


def process_full_file_batched(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options=None, config=None, measure=False):
    """process_full_file with the per-region work of ALL regions done in a few launches (sono_analyze_rois: ragged
    STFT batch, dB crop, per-frame peaks, noise-floor histograms, high-pass, all in HBM) and one host pass
    (sono_frame_roi_calls); with measure=True the measurement stage of all calls follows in one more launch and a
    pool of host threads (sono_measure_roi_calls).  Same calls, same order, same values as process_full_file."""
    opt = dict({"threshold_dB": -60, "padding_ms": 5}, **(options or {}))
    cfg = dict(DEFAULT_FRAMING_CONFIG, **(config or {}))
    x = _f32(full_audio_data)
    raw = _det.fast_scan_segments(engine, x, sample_rate, flow_khz, fhigh_khz, opt["threshold_dB"])          # js:594
    if not raw:
        return []
    merged = _det.merge_and_pad_segments(raw, x.size, sample_rate, opt["padding_ms"])                        # js:601
    n = len(merged)
    rois = (_Segment * n)(*[_Segment(m["startSample"], m["endSample"]) for m in merged])
    u64p, i32p, dp = C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    row0 = np.zeros(n + 1, np.uint64)
    n_fft = engine.get_fft_size()
    _check(lib().sono_roi_rows(rois, n, n_fft, float(cfg["hopPercent"]), row0.ctypes.data_as(u64p)))
    total = int(row0[-1])
    peak_db = np.zeros(max(total, 1), np.float32)
    peak_bin = np.zeros(max(total, 1), np.int32)
    zonal = np.zeros(n * SONO_MAX_ZONES, np.float64)
    hp = np.zeros(n, np.float32)
    left, right = np.zeros_like(peak_db), np.zeros_like(peak_db)
    ilo, ihi = np.zeros_like(peak_bin), np.zeros_like(peak_bin)
    table = _RoiTable(row0.ctypes.data_as(u64p), _p(peak_db), peak_bin.ctypes.data_as(i32p), _p(left), _p(right),
                      ilo.ctypes.data_as(i32p), ihi.ctypes.data_as(i32p), zonal.ctypes.data_as(dp), _p(hp), 0, 0, 0, 0)
    xf = np.empty_like(x)                                 # receives the high-passed regions, nothing else is touched
    _check(lib().sono_analyze_rois(engine._handle(), _p(x), x.size, rois, n, float(sample_rate), float(flow_khz), float(fhigh_khz),
                                   float(cfg["hopPercent"]), 1, C.byref(table), total, _p(xf)))
    cap = total // 2 + n + 1
    out = (_RoiCall * cap)()
    got = C.c_size_t(0)
    _check(lib().sono_frame_roi_calls(_p(x), _p(xf), x.size, rois, n, C.byref(table), float(sample_rate), n_fft,
                                      float(cfg["callThreshold_dB"]), float(cfg["minCallDuration_ms"]), out, cap, C.byref(got)))
    calls = []
    if measure:
        from . import CallMeasurement as _Measurement, measure as _ms
        m = (_Measurement * max(got.value, 1))()
        _check(lib().sono_measure_roi_calls(engine._handle(), _p(x), _p(xf), x.size, rois, n, C.byref(table), out, got.value,
                                            float(sample_rate), float(flow_khz), float(fhigh_khz), float(cfg["hopPercent"]),
                                            _ms.mode_flags(cfg.get("debugMode", False), cfg.get("timeExpansion", False)), 0, m))
        for i in range(got.value):
            c, r = out[i], int(out[i].roi)
            if m[i].reference_throws:
                raise _ms.ReferenceThrows("measureFrequencyParameters")
            call = {name: (None if v != v else v) for name, v in ((nm, getattr(m[i], f)) for nm, (f, _) in
                                                                  zip(_ms.CALL_FIELDS, _Measurement._fields_))}
            call.update({name: (None if getattr(m[i], f) < 0 else int(getattr(m[i], f)))
                         for name, f in zip(_ms._FRAME_FIELDS, _Measurement._frames)})
            if m[i].discarded:
                call["isDiscarded"] = True
            zmap = {10 * z: float(zonal[r * SONO_MAX_ZONES + z]) for z in range(table.num_zones)}
            call = _ms.finish_full_file_call(call, merged[r]["startSample"] / sample_rate)
            if call is not None:
                calls.append(dict(call, startFrame=int(c.start_frame), endFrame=int(c.end_frame), zonalNoiseMap=zmap, roi=r,
                                  highpass_kHz=float(hp[r])))
        return calls
    for i in range(got.value):
        c = out[i]
        if c.discarded:                                      # a vertical click: the reference drops the call (js:802)
            continue
        r = int(c.roi)
        calls.append({
            "roi": r, "startFrame": int(c.start_frame), "endFrame": int(c.end_frame), "startTime_s": c.start_time_s,
            "endTime_s": c.end_time_s, "duration_ms": c.duration_ms, "peakPower_dB": float(c.peak_power_db),
            "peakTime_s": c.peak_time_s, "peakFreq_kHz": c.peak_freq_khz, "peakFreq_ms": c.peak_freq_ms,
            "instBandwidth_kHz": c.inst_bandwidth_khz, "snr_dB": float(c.peak_power_db) + 80.0, "highpass_kHz": float(hp[r]),
            "zonalNoiseMap": {10 * z: float(zonal[r * SONO_MAX_ZONES + z]) for z in range(table.num_zones)},
        })
    return calls


_CALL_TYPES = ("CF", "FM", "CF-FM")
_QUALITIES = ("Very Poor", "Poor", "Normal", "Good", "Excellent")


def _full_file_call_array(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options, config, threads, keep_spectrograms,
                          pcm_channel=0):
    """sono_process_full_file -> (ctypes array of sono_full_file_call, number of calls).  An int16 array -- one
    dimension: a mono recording; two: sample frames x channels, of which pcm_channel is analysed -- goes through
    sono_process_full_file_pcm16 as it is (the samples mean value / 32768, as a decoded 16-bit WAV file's do)."""
    from . import FullFileCall as _Call, FullFileOptions as _Opt, SonoError, measure as _ms
    opt = dict({"threshold_dB": -60, "padding_ms": 5}, **(options or {}))
    cfg = dict(DEFAULT_FRAMING_CONFIG, **(config or {}))
    pcm = isinstance(full_audio_data, np.ndarray) and full_audio_data.dtype == np.int16
    x = np.ascontiguousarray(full_audio_data) if pcm else _f32(full_audio_data)
    if pcm and x.ndim not in (1, 2):
        raise ValueError("16-bit recordings are (frames,) or (frames, channels) arrays")
    o = _Opt(float(opt["threshold_dB"]), float(opt["padding_ms"]), float(cfg["hopPercent"]), float(cfg["callThreshold_dB"]),
             float(cfg["minCallDuration_ms"]), _ms.mode_flags(cfg.get("debugMode", False), cfg.get("timeExpansion", False)) | (4 if keep_spectrograms else 0),
             int(threads))
    cap = 4096
    out = (_Call * cap)()
    got = C.c_size_t(0)
    try:
        if pcm:
            _check(lib().sono_process_full_file_pcm16(engine._handle(), x.ctypes.data, x.shape[0], 1 if x.ndim == 1 else x.shape[1],
                                                      int(pcm_channel), float(sample_rate), float(flow_khz), float(fhigh_khz),
                                                      C.byref(o), out, cap, C.byref(got)))
        else:
            _check(lib().sono_process_full_file(engine._handle(), _p(x), x.size, float(sample_rate), float(flow_khz), float(fhigh_khz),
                                                C.byref(o), out, cap, C.byref(got)))
    except SonoError as err:
        if err.status == 5:                                # SONO_EREFERENCE
            raise _ms.ReferenceThrows(str(err)) from None
        if err.status != 4:                                # SONO_ECAPACITY: the engine kept the calls, fetch them all
            raise
        cap = got.value
        out = (_Call * cap)()
        _check(lib().sono_full_file_results(engine._handle(), out, cap, C.byref(got)))
    return out, got.value


# sharding.COLUMNS -> field of sono_full_file_call ("m." = inside its sono_call_measurement)
_TABLE_FIELDS = {"startTime_s": "m.start_time_s", "endTime_s": "m.end_time_s", "duration_ms": "m.duration_ms",
                 "bandwidth_kHz": "m.bandwidth_khz", "startFreq_kHz": "m.start_freq_khz", "startFreq_ms": "m.start_freq_ms",
                 "endFreq_kHz": "m.end_freq_khz", "endFreq_ms": "m.end_freq_ms", "highFreq_kHz": "m.high_freq_khz",
                 "highFreq_ms": "m.high_freq_ms", "lowFreq_kHz": "m.low_freq_khz", "lowFreq_ms": "m.low_freq_ms",
                 "peakFreq_kHz": "m.peak_freq_khz", "peakFreq_ms": "m.peak_freq_ms",
                 "characteristicFreq_kHz": "m.characteristic_freq_khz", "characteristicFreq_ms": "m.characteristic_freq_ms",
                 "kneeFreq_kHz": "m.knee_freq_khz", "kneeFreq_ms": "m.knee_freq_ms", "heelFreq_kHz": "m.heel_freq_khz",
                 "heelFreq_ms": "m.heel_freq_ms", "peakPower_dB": "m.peak_power_db", "snr_dB": "snr_db"}


def process_full_file_table(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options=None, config=None, threads=0,
                            file_id=0, channel=0):
    """process_full_file_native straight into the call-summary table's rows (sharding.COLUMNS order, float64, NaN where
    the reference leaves null): equal to sharding.calls_to_rows(process_full_file_native(...), file_id, channel) without
    a Python object per call -- what a survey over many recordings wants."""
    from . import FullFileCall as _Call, sharding as _sh
    out, count = _full_file_call_array(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options, config, threads, False)
    rows = np.full((count, len(_sh.COLUMNS)), np.nan, np.float64)
    if count == 0:
        return rows
    rec = np.frombuffer(out, dtype=np.dtype(_Call), count=count)
    rows[:, 0], rows[:, 1] = file_id, channel
    for j, name in enumerate(_sh.COLUMNS[2:], start=2):
        field = _TABLE_FIELDS[name]
        rows[:, j] = rec["m"][field[2:]] if field.startswith("m.") else rec[field]
    return rows


def process_full_file_native(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options=None, config=None, threads=0,
                             keep_spectrograms=False):
    """processFullFile (batCallDetector.js:584-840) as ONE library call (sono_process_full_file): same calls, order and
    values as process_full_file_batched(..., measure=True), without the per-region tables (zonalNoiseMap).  Raises
    measure.ReferenceThrows where the reference's promise rejects (config['debugMode'] False and a call on which the
    reference raises outside its debug mode).  keep_spectrograms=True adds what the reference leaves on every BatCall
    beside the numbers (js:785-787): 'spectrogram' (the call's rows of its region's dB matrix), 'timeFrames' (their centre
    times from the region's start, and one more) and 'freqBins' (Hz)."""
    from . import measure as _ms
    out, count = _full_file_call_array(engine, full_audio_data, sample_rate, flow_khz, fhigh_khz, options, config, threads,
                                       keep_spectrograms)
    calls = []
    for i in range(count):
        k, m = out[i], out[i].m
        call = {name: (None if v != v else v) for name, v in ((nm, getattr(m, f)) for nm, (f, _) in zip(_ms.CALL_FIELDS, type(m)._fields_))}
        call.update({name: (None if getattr(m, f) < 0 else int(getattr(m, f))) for name, f in zip(_ms._FRAME_FIELDS, type(m)._frames)})
        call.update(Flow=k.flow_hz, Fhigh=k.fhigh_khz, callType=_CALL_TYPES[k.call_type], snr_dB=k.snr_db,
                    quality=_QUALITIES[k.quality], startFrame=int(k.start_frame), endFrame=int(k.end_frame), roi=int(k.roi),
                    highpass_kHz=float(k.highpass_khz))
        if keep_spectrograms:
            frames, times, nb = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
            args = (engine._handle(), i)
            _check(lib().sono_full_file_call_spectrogram(*args, None, 0, None, 0, None, 0, C.byref(frames), C.byref(times), C.byref(nb)))
            db = np.empty((frames.value, nb.value), np.float32)
            tf, fb = np.empty(times.value, np.float64), np.empty(nb.value, np.float32)
            _check(lib().sono_full_file_call_spectrogram(*args, _p(db), db.size, tf.ctypes.data_as(C.POINTER(C.c_double)), tf.size,
                                                         _p(fb), fb.size, None, None, None))
            call.update(spectrogram=db, timeFrames=tf, freqBins=fb)
        calls.append(call)
    return calls


def calculate_rms_based_snr(engine, call, spectrogram, freq_bins, signal_start_idx, signal_end_idx, flow_khz, fhigh_khz,
                            noise_spectrogram=None):
    """calculateRMSbasedSNR (batCallDetector.js:402-571).  call: {'lowFreq_kHz', 'highFreq_kHz'}; noise_spectrogram:
    generate_spectrogram's dict or None.  -> {'snr_dB' (None where the reference leaves null), 'mechanism',
    'signalPowerMean_dB', 'noisePowerMean_dB', 'signalCount', 'noiseCount', 'signalThreshold', 'signalMax', 'signalMin'}."""
    p = np.ascontiguousarray(spectrogram, dtype=np.float32)
    fb = _f32(freq_bins)
    F, B = (p.shape if p.ndim == 2 else (0, fb.size))
    ext = noise_spectrogram is not None and len(noise_spectrogram["powerMatrix"]) > 0
    if ext:
        q = np.ascontiguousarray(noise_spectrogram["powerMatrix"], dtype=np.float32)
        nfb = _f32(noise_spectrogram["freqBins"])
    r = _SnrResult()
    _check(lib().sono_rms_snr(engine._handle(), _p(p) if p.size else None, F, B, _p(fb), int(signal_start_idx), int(signal_end_idx),
                              float(call["lowFreq_kHz"]) * 1000, float(call["highFreq_kHz"]) * 1000,
                              _p(q) if ext else None, q.shape[0] if ext else 0, q.shape[1] if ext else 0, _p(nfb) if ext else None,
                              float(flow_khz) * 1000, float(fhigh_khz) * 1000, C.byref(r)))
    nn = lambda v: None if v != v else v
    return {"snr_dB": r.snr_db if r.status == 0 else None,
            "mechanism": "RMS-based (Last 10ms)" if ext else "RMS-based (Fallback Internal)",
            "signalPowerMean_dB": nn(r.signal_mean_db), "noisePowerMean_dB": nn(r.noise_mean_db),
            "signalCount": int(r.signal_count), "noiseCount": int(r.noise_count), "signalThreshold": nn(r.threshold_db),
            "signalMax": nn(r.signal_max_db), "signalMin": nn(r.signal_min_db)}
