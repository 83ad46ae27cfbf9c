"""The measurement stage of the reference's BatCallDetector (modules/batCallDetector.js), as far as it is built: the two
threshold searches, under the reference's method names (snake_case) and return shapes.  Serial per call; host C++ in
libsonostft (csrc/sono_measure.cpp), needs no GPU.
"""
import ctypes as C

import numpy as np

from . import CallMatrix as _CallMatrix, HighThreshold as _High, LowThreshold as _Low, _check, _f32, _p, lib

__all__ = ["find_optimal_low_frequency_threshold", "find_optimal_high_frequency_threshold", "ReferenceThrows"]


class ReferenceThrows(RuntimeError):
    """The reference, outside its debug mode, raises a TypeError on these inputs (it consults a log table that only its
    debug mode creates: batCallDetector.js:2815, 2261)."""


def _matrix(spectrogram, time_frames, freq_bins, zonal_noise_map):
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
                    z.ctypes.data_as(dp) if z is not None else None, 0 if z is None else z.size)
    return m, (p, fb, tf, z)


def find_optimal_low_frequency_threshold(spectrogram, time_frames, freq_bins, flow_khz, fhigh_khz, call_peak_power_db,
                                         peak_frame_idx=0, limit_frame_idx=None, zonal_noise_map=None):
    """findOptimalLowFrequencyThreshold (batCallDetector.js:2413-2909) -> the reference's result dict (None for null)."""
    m, keep = _matrix(spectrogram, time_frames, freq_bins, zonal_noise_map)
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
                                          peak_frame_idx=0, zonal_noise_map=None):
    """findOptimalHighFrequencyThreshold (batCallDetector.js:1931-2401) -> the reference's result dict (None for null)."""
    m, keep = _matrix(spectrogram, time_frames, freq_bins, zonal_noise_map)
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
