"""The measurement stage of the reference's BatCallDetector (modules/batCallDetector.js): measureFrequencyParameters and
its two threshold searches, under the reference's method names (snake_case) and return shapes.  Serial per call; host
C++ in libsonostft (csrc/sono_measure.cpp), needs no GPU.
"""
import ctypes as C

import numpy as np

from . import CallMatrix as _CallMatrix, CallMeasurement as _Measurement, HighThreshold as _High, LowThreshold as _Low
from . import _check, _f32, _p, lib

__all__ = ["find_optimal_low_frequency_threshold", "find_optimal_high_frequency_threshold", "measure_frequency_parameters",
           "classify_call_type", "get_quality_rating", "finish_full_file_call", "ReferenceThrows", "CALL_FIELDS"]


class ReferenceThrows(RuntimeError):
    """The reference, outside its debug mode, raises a TypeError on these inputs (it consults a log table that only its
    debug mode creates: batCallDetector.js:2815, 2261)."""


def mode_flags(debug_mode=False, time_expansion=False):
    """The library's `debug_mode` word: SONO_DEBUG_MODE (the detector's debugMode, batCallDetector.js:359-361) and
    SONO_TIME_EXPANSION (getTimeExpansionMode(), fileState.js:158-166) or-ed together."""
    return (1 if debug_mode else 0) | (2 if time_expansion else 0)


def _matrix(spectrogram, time_frames, freq_bins, zonal_noise_map, debug_mode=False, time_expansion=False):
    p = np.ascontiguousarray(spectrogram, dtype=np.float32)
    fb = _f32(freq_bins)
    tf = np.ascontiguousarray(time_frames, dtype=np.float64)
    F, B = (p.shape if p.ndim == 2 else (0, fb.size))
    dp = C.POINTER(C.c_double)
    z = None
    if zonal_noise_map is not None:
        keys = sorted(int(k) for k in zonal_noise_map)
        z = np.asarray([zonal_noise_map[k] if k in zonal_noise_map else zonal_noise_map[str(k)] for k in keys], np.float64)
        if keys != [10 * i for i in range(len(keys))]:
            raise ValueError("zonal noise map keys must be 0, 10, 20, ...")
    m = _CallMatrix(_p(p) if p.size else None, F, B, _p(fb), tf.ctypes.data_as(dp) if tf.size else None,
                    z.ctypes.data_as(dp) if z is not None else None, 0 if z is None else z.size, mode_flags(debug_mode, time_expansion), 0)
    return m, (p, fb, tf, z)


def find_optimal_low_frequency_threshold(spectrogram, time_frames, freq_bins, flow_khz, fhigh_khz, call_peak_power_db,
                                         peak_frame_idx=0, limit_frame_idx=None, zonal_noise_map=None, debug_mode=False):
    """findOptimalLowFrequencyThreshold (batCallDetector.js:2413-2909) -> the reference's result dict (None for null).
    debug_mode = the detector's debugMode switch: outside it the reference raises on some inputs (ReferenceThrows)."""
    m, keep = _matrix(spectrogram, time_frames, freq_bins, zonal_noise_map, debug_mode)
    r = _Low()
    _check(lib().sono_find_optimal_low_frequency_threshold(C.byref(m), float(call_peak_power_db), int(peak_frame_idx),
                                                           -1 if limit_frame_idx is None else int(limit_frame_idx), C.byref(r)))
    if r.reference_throws:
        raise ReferenceThrows("findOptimalLowFrequencyThreshold")
    lo = r.low_freq_hz if r.has_freq else None
    en = r.end_freq_hz if r.has_freq else None
    out = {"threshold": r.threshold_db, "lowFreq_Hz": lo, "lowFreq_kHz": None if lo is None else lo / 1000, "endFreq_Hz": en,
           "endFreq_kHz": None if en is None else en / 1000, "warning": bool(r.warning)}
    if r.discard:
        out.update(lowFreqFrameIdx=None, discard=True)
    elif r.has_frame:
        out.update(lowFreqFrameIdx=int(r.low_freq_frame), lowFreqBinIdx=int(r.low_freq_bin))
    return out


def find_optimal_high_frequency_threshold(spectrogram, time_frames, freq_bins, flow_khz, fhigh_khz, call_peak_power_db,
                                          peak_frame_idx=0, zonal_noise_map=None, debug_mode=False):
    """findOptimalHighFrequencyThreshold (batCallDetector.js:1931-2401) -> the reference's result dict (None for null)."""
    m, keep = _matrix(spectrogram, time_frames, freq_bins, zonal_noise_map, debug_mode)
    r = _High()
    _check(lib().sono_find_optimal_high_frequency_threshold(C.byref(m), float(call_peak_power_db), int(peak_frame_idx), C.byref(r)))
    if r.reference_throws:
        raise ReferenceThrows("findOptimalHighFrequencyThreshold")
    hi = r.high_freq_hz if r.has_freq else None
    st = r.start_freq_hz if r.has_start else None
    out = {"threshold": r.threshold_db, "highFreq_Hz": hi, "highFreq_kHz": None if hi is None else hi / 1000,
           "highFreqFrameIdx": int(r.high_freq_frame), "startFreq_Hz": st, "startFreq_kHz": None if st is None else st / 1000,
           "warning": bool(r.warning)}
    if r.has_limit:
        out.update(highFreqBinIdx=int(r.high_freq_bin), finalSearchLimitFrame=int(r.final_search_limit_frame),
                   isCFStablePattern=bool(r.is_cf_stable_pattern))
    return out


# BatCall field names (batCallDetector.js:76-145) of what the measurement writes, in the order of struct sono_call_measurement.
CALL_FIELDS = ("startTime_s", "endTime_s", "duration_ms", "peakFreq_kHz", "peakFreq_ms", "peakPower_dB", "highFreq_kHz", "highFreq_ms",
               "startFreq_kHz", "startFreq_ms", "startFreqTime_s", "endFreq_kHz", "endFreq_ms", "endFreqTime_s", "lowFreq_kHz",
               "lowFreq_ms", "characteristicFreq_kHz", "characteristicFreq_ms", "kneeFreq_kHz", "kneeFreq_ms", "heelFreq_kHz",
               "heelFreq_ms", "bandwidth_kHz", "highFreqThreshold_dB_used", "lowFreqThreshold_dB_used")
_FRAME_FIELDS = ("peakFrameIdx", "highFreqFrameIdx", "startFreqFrameIdx", "startFreqBinIdx", "endFrameIdx_forLowFreq", "kneeFrameIdx",
                 "heelFrameIdx")


def measure_frequency_parameters(call, flow_khz, fhigh_khz, freq_bins, freq_resolution, zonal_noise_map=None, debug_mode=False,
                                 time_expansion=False):
    """measureFrequencyParameters (batCallDetector.js:2994-3894).  `call` is a dict with the reference's BatCall fields
    "spectrogram" (frames x bins, dB), "timeFrames" (the frames' centre times, one more than frames unless the call ends
    the matrix), "startTime_s", "endTime_s", "duration_ms"; it is updated in place like the reference's object (None for
    null; "isDiscarded" set when the reference sets it) and returned.  Raises ReferenceThrows where the reference does.
    time_expansion = getTimeExpansionMode() (fileState.js:158-166): a call that reaches the end of the measurement gets
    BatCall.applyTimeExpansion(10) (js:185-278, 3890-3893), frequencies x 10, times and durations / 10."""
    tf = np.ascontiguousarray(call["timeFrames"], dtype=np.float64)
    m, keep = _matrix(call["spectrogram"], tf, freq_bins, zonal_noise_map, debug_mode, time_expansion)
    r = _Measurement()
    nan = float("nan")
    r.start_time_s, r.end_time_s, r.duration_ms = (nan if call.get(k) is None else float(call[k])
                                                   for k in ("startTime_s", "endTime_s", "duration_ms"))
    _check(lib().sono_measure_frequency_parameters(C.byref(m), tf.size, float(fhigh_khz), float(freq_resolution), C.byref(r)))
    for name, (field, _) in zip(CALL_FIELDS, _Measurement._fields_):
        v = getattr(r, field)
        call[name] = None if v != v else v
    for name, field in zip(_FRAME_FIELDS, _Measurement._frames):
        v = getattr(r, field)
        call[name] = None if v < 0 else int(v)
    if r.reference_throws:
        raise ReferenceThrows("measureFrequencyParameters")
    if r.discarded:
        call["isDiscarded"] = True
    return call


def classify_call_type(call):
    """CallTypeClassifier.classify (batCallDetector.js:48-56)."""
    bw = call.get("bandwidth_kHz")
    if not bw or bw != bw or bw < 5:
        return "CF"
    return "FM" if bw > 20 else "CF-FM"


def get_quality_rating(snr_db):
    """getQualityRating (batCallDetector.js:375-387)."""
    for limit, name in ((10, "Very Poor"), (15, "Poor"), (20, "Normal"), (30, "Good")):
        if snr_db < limit:
            return name
    return "Excellent"


def finish_full_file_call(call, time_offset_s):
    """What processFullFile does with a measured call (batCallDetector.js:802-825): None if it was discarded or lasts
    1 ms or less; else classification, the placeholder SNR (peak + 80 dB), quality, and region time -> file time."""
    if call.get("isDiscarded") or call["duration_ms"] <= 1.0:
        return None
    call["Flow"] = call["lowFreq_kHz"] * 1000
    call["Fhigh"] = call["highFreq_kHz"]
    call["callType"] = classify_call_type(call)
    call["snr_dB"] = call["peakPower_dB"] - (-80)
    call["quality"] = get_quality_rating(call["snr_dB"])
    for k in ("startTime_s", "endTime_s", "startFreqTime_s", "endFreqTime_s"):
        if call.get(k) is not None:
            call[k] += time_offset_s
    return call
