"""spectrogram_amd -- Python host side of libsonostft (MI355X spectrogram hot path).

The classes and functions here carry the names, argument meaning and error
behaviour of the reference's wasm export surface
(modules/spectrogram_wasm.d.ts:4-271), so code written against
`SpectrogramEngine`, `WaveformEngine`, `compute_power_spectrum`, ... of the wasm
module reads the same against this package.  Every numeric call goes through the
C ABI of include/sonostft.h into HIP kernels; there is no CPU compute path, and
a missing library or GPU raises instead of falling back.
"""
import ctypes as C
import os

import numpy as np

__all__ = [
    "SpectrogramEngine", "WaveformEngine", "compute_power_spectrum",
    "find_peak_frequency_from_spectrum", "compute_wave_peaks", "find_global_max", "wav_info", "decode_wav",
    "frames_for", "device_count", "lib", "LIB_PATH", "SonoError",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsonostft.so")

SONO_OK, SONO_EINVAL, SONO_ENOMEM, SONO_EDEVICE, SONO_ECAPACITY = 0, 1, 2, 3, 4

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_szp = C.POINTER(C.c_size_t)


class SonoError(RuntimeError):
    """A non-zero status from libsonostft (the wasm module throws a JS Error)."""

    def __init__(self, status, message):
        super().__init__(f"libsonostft status {status}: {message}")
        self.status = status


class Batch(C.Structure):
    """struct sono_batch of include/sonostft.h."""
    _fields_ = [("d_samples", C.c_void_p), ("n_streams", C.c_size_t),
                ("stream_stride", C.c_size_t), ("samples_per_stream", C.c_size_t)]


class RaggedBatch(C.Structure):
    """struct sono_ragged_batch."""
    _fields_ = [("d_samples", C.c_void_p), ("n_streams", C.c_size_t), ("offsets", C.POINTER(C.c_uint64)),
                ("lengths", C.POINTER(C.c_uint64))]


class Segment(C.Structure):
    """struct sono_segment."""
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64)]


class RoiTable(C.Structure):
    """struct sono_roi_table."""
    _fields_ = [("row0", C.POINTER(C.c_uint64)), ("peak_db", C.POINTER(C.c_float)), ("peak_bin", C.POINTER(C.c_int32)),
                ("peak_left_db", C.POINTER(C.c_float)), ("peak_right_db", C.POINTER(C.c_float)),
                ("inst_lo_bin", C.POINTER(C.c_int32)), ("inst_hi_bin", C.POINTER(C.c_int32)), ("zonal", C.POINTER(C.c_double)), ("highpass_khz", C.POINTER(C.c_float)), ("num_zones", C.c_uint32),
                ("min_bin", C.c_uint32), ("num_bins", C.c_uint32), ("hop", C.c_uint32)]


class RoiCall(C.Structure):
    """struct sono_roi_call."""
    _fields_ = [("roi", C.c_uint64), ("start_frame", C.c_uint64), ("end_frame", C.c_uint64), ("peak_frame", C.c_uint64),
                ("start_time_s", C.c_double), ("end_time_s", C.c_double), ("duration_ms", C.c_double),
                ("peak_time_s", C.c_double), ("peak_freq_khz", C.c_double), ("peak_freq_ms", C.c_double),
                ("inst_bandwidth_khz", C.c_double), ("peak_power_db", C.c_float), ("peak_bin", C.c_int32),
                ("discarded", C.c_int32), ("pad", C.c_int32)]


SONO_MAX_ZONES = 32


class CallMatrix(C.Structure):
    """struct sono_call_matrix."""
    _fields_ = [("power_db", C.POINTER(C.c_float)), ("frames", C.c_size_t), ("num_bins", C.c_uint32),
                ("freq_bins", C.POINTER(C.c_float)), ("time_frames", C.POINTER(C.c_double)), ("zonal", C.POINTER(C.c_double)),
                ("num_zones", C.c_size_t), ("debug_mode", C.c_int32), ("pad", C.c_int32)]


class CallMeasurement(C.Structure):
    """struct sono_call_measurement."""
    _doubles = ("start_time_s", "end_time_s", "duration_ms", "peak_freq_khz", "peak_freq_ms", "peak_power_db", "high_freq_khz",
                "high_freq_ms", "start_freq_khz", "start_freq_ms", "start_freq_time_s", "end_freq_khz", "end_freq_ms",
                "end_freq_time_s", "low_freq_khz", "low_freq_ms", "characteristic_freq_khz", "characteristic_freq_ms",
                "knee_freq_khz", "knee_freq_ms", "heel_freq_khz", "heel_freq_ms", "bandwidth_khz", "high_threshold_db_used",
                "low_threshold_db_used")
    _frames = ("peak_frame", "high_freq_frame", "start_freq_frame", "start_freq_bin", "end_frame_for_low_freq", "knee_frame",
               "heel_frame")
    _fields_ = ([(n, C.c_double) for n in _doubles] + [(n, C.c_int64) for n in _frames]
                + [("discarded", C.c_int32), ("reference_throws", C.c_int32)])


class FullFileOptions(C.Structure):
    """struct sono_full_file_options."""
    _fields_ = [("threshold_db", C.c_double), ("padding_ms", C.c_double), ("hop_percent", C.c_double),
                ("call_threshold_db", C.c_double), ("min_call_duration_ms", C.c_double), ("debug_mode", C.c_int32),
                ("threads", C.c_uint32)]


class FullFileCall(C.Structure):
    """struct sono_full_file_call."""
    _fields_ = [("m", CallMeasurement), ("roi", C.c_uint64), ("start_frame", C.c_uint64), ("end_frame", C.c_uint64),
                ("highpass_khz", C.c_double), ("flow_hz", C.c_double), ("fhigh_khz", C.c_double), ("snr_db", C.c_double),
                ("call_type", C.c_int32), ("quality", C.c_int32)]


class LowThreshold(C.Structure):
    """struct sono_low_threshold."""
    _fields_ = [("threshold_db", C.c_double), ("low_freq_hz", C.c_double), ("end_freq_hz", C.c_double),
                ("low_freq_frame", C.c_int64), ("low_freq_bin", C.c_int64), ("has_freq", C.c_int32), ("has_frame", C.c_int32),
                ("warning", C.c_int32), ("discard", C.c_int32), ("reference_throws", C.c_int32), ("pad", C.c_int32)]


class HighThreshold(C.Structure):
    """struct sono_high_threshold."""
    _fields_ = [("threshold_db", C.c_double), ("high_freq_hz", C.c_double), ("start_freq_hz", C.c_double),
                ("high_freq_bin", C.c_int64), ("high_freq_frame", C.c_int64), ("final_search_limit_frame", C.c_int64),
                ("has_freq", C.c_int32), ("has_start", C.c_int32), ("has_limit", C.c_int32), ("warning", C.c_int32),
                ("is_cf_stable_pattern", C.c_int32), ("reference_throws", C.c_int32)]


class SnrResult(C.Structure):
    """struct sono_snr_result."""
    _fields_ = [("snr_db", C.c_double), ("signal_mean_db", C.c_double), ("noise_mean_db", C.c_double),
                ("threshold_db", C.c_double), ("signal_max_db", C.c_double), ("signal_min_db", C.c_double),
                ("signal_count", C.c_uint64), ("noise_count", C.c_uint64), ("status", C.c_int32), ("pad", C.c_int32)]


class CallFrame(C.Structure):
    """struct sono_call_frame."""
    _fields_ = [("start_frame", C.c_uint64), ("end_frame", C.c_uint64), ("start_time_s", C.c_double),
                ("end_time_s", C.c_double), ("duration_ms", C.c_double)]


class FrameFeatures(C.Structure):
    """struct sono_frame_features."""
    _fields_ = [("peak_bin", C.c_int32), ("peak_db", C.c_float), ("low_bin", C.c_int32), ("high_bin", C.c_int32)]


class WavInfo(C.Structure):
    """struct sono_wav_info."""
    _fields_ = [("format_tag", C.c_uint32), ("channels", C.c_uint32), ("sample_rate", C.c_uint32),
                ("bits_per_sample", C.c_uint32), ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64),
                ("frames", C.c_uint64), ("guano_offset", C.c_uint64), ("guano_bytes", C.c_uint64)]


_segp = C.POINTER(Segment)
_ffp = C.POINTER(FrameFeatures)

_lib = None

_SIGNATURES = {
    "sono_last_error": (C.c_char_p, []),
    "sono_device_count": (C.c_int, []),
    "sono_abi_version": (C.c_uint32, []),
    "sono_frames_for": (C.c_size_t, [C.c_size_t, C.c_uint32, C.c_uint32]),
    "sono_engine_create": (C.c_int, [C.c_uint32, C.c_char_p, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_void_p)]),
    "sono_engine_destroy": (None, [C.c_void_p]),
    "sono_engine_fft_size": (C.c_uint32, [C.c_void_p]),
    "sono_engine_freq_bins": (C.c_uint32, [C.c_void_p]),
    "sono_engine_num_filters": (C.c_uint32, [C.c_void_p]),
    "sono_engine_device": (C.c_int, [C.c_void_p]),
    "sono_set_host_threads": (C.c_int, [C.c_uint32]),
    "sono_host_threads": (C.c_uint32, []),
    "sono_engine_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "sono_engine_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "sono_engine_window_values": (C.c_int, [C.c_void_p, _f32p, C.c_size_t]),
    "sono_engine_load_filter_bank": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32]),
    "sono_engine_clear_filter_bank": (C.c_int, [C.c_void_p]),
    "sono_engine_release_memory": (C.c_int, [C.c_void_p]),
    "sono_engine_synchronize": (C.c_int, [C.c_void_p]),
    "sono_stft_mag_f32": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, _f32p, C.c_size_t, _szp]),
    "sono_stft_u8": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, C.c_float, C.c_float, _u8p,
                               C.c_size_t, _szp]),
    "sono_peaks_u16": (C.c_int, [C.c_void_p, C.c_float, _u16p, C.c_size_t, _szp]),
    "sono_peak_mags_f32": (C.c_int, [C.c_void_p, C.c_float, _f32p, C.c_size_t, _szp]),
    "sono_global_max": (C.c_float, [C.c_void_p]),
    "sono_engine_set_color_map": (C.c_int, [C.c_void_p, _u8p, C.c_size_t]),
    "sono_engine_set_spectrum_config": (C.c_int, [C.c_void_p, C.c_char_p, C.c_float, C.c_float]),
    "sono_spectrogram_image": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_float, C.c_float, _u8p, C.c_size_t]),
    "sono_stft_u8_dev": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_float, C.c_float,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_stft_u8_pcm16_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "sono_stft_band_active_pcm16_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32,
                                                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]),
    "sono_stft_survey_dev": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_stft_survey_pcm16_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "sono_stft_mag_f32_dev": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_void_p, C.c_void_p]),
    "sono_stft_features_dev": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "sono_fast_scan_segments": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                                          _segp, C.c_size_t, _szp]),
    "sono_generate_spectrogram": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                            C.c_double, _f32p, C.c_size_t, _szp, _u32p, _u32p]),
    "sono_detect_call_segments": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, C.c_double, _segp, C.c_size_t,
                                            _szp]),
    "sono_frame_features_db": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_double, C.c_uint32, C.c_double, _ffp, C.c_size_t]),
    "sono_merge_and_pad_segments": (C.c_int, [_segp, C.c_size_t, C.c_uint64, C.c_double, C.c_double, _segp, C.c_size_t,
                                              _szp]),
    "sono_ragged_rows": (C.c_int, [C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "sono_stft_u8_ragged_dev": (C.c_int, [C.c_void_p, C.POINTER(RaggedBatch), C.c_uint32, C.c_float, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_stft_mag_f32_ragged_dev": (C.c_int, [C.c_void_p, C.POINTER(RaggedBatch), C.c_uint32, C.c_void_p, C.c_void_p]),
    "sono_stft_band_active_dev": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_detect_band_active_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_detect_power_db_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p]),
    "sono_detect_frame_peaks_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_int, C.c_void_p]),
    "sono_detect_db_max_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sono_detect_db_active_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_double, C.c_void_p,
                                            C.c_void_p]),
    "sono_detect_frame_features_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_double, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p]),
    "sono_detect_runs_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "sono_detect_runs_static_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "sono_zonal_noise_floors": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, _f32p, C.c_size_t, C.c_size_t,
                                          C.POINTER(C.c_double), C.c_size_t, _szp]),
    "sono_zonal_noise_floors_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, _f32p, C.c_size_t, C.c_size_t,
                                              C.POINTER(C.c_double), C.c_size_t, _szp, C.c_void_p]),
    "sono_db_percentile": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_double, C.POINTER(C.c_float)]),
    "sono_db_percentile_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.POINTER(C.c_float), C.c_void_p]),
    "sono_process_full_file": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                         C.POINTER(FullFileOptions), C.POINTER(FullFileCall), C.c_size_t, _szp]),
    "sono_process_full_file_pcm16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                               C.c_double, C.POINTER(FullFileOptions), C.POINTER(FullFileCall), C.c_size_t, _szp]),
    "sono_full_file_results": (C.c_int, [C.c_void_p, C.POINTER(FullFileCall), C.c_size_t, _szp]),
    "sono_full_file_call_spectrogram": (C.c_int, [C.c_void_p, C.c_size_t, _f32p, C.c_size_t, C.POINTER(C.c_double), C.c_size_t, _f32p,
                                                  C.c_size_t, _szp, _szp, _u32p]),
    "sono_measure_roi_calls": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_size_t, C.POINTER(Segment), C.c_size_t, C.POINTER(RoiTable),
                                         C.POINTER(RoiCall), C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                         C.c_uint, C.POINTER(CallMeasurement)]),
    "sono_measure_frequency_parameters": (C.c_int, [C.POINTER(CallMatrix), C.c_size_t, C.c_double, C.c_double,
                                                    C.POINTER(CallMeasurement)]),
    "sono_find_optimal_low_frequency_threshold": (C.c_int, [C.POINTER(CallMatrix), C.c_double, C.c_int64, C.c_int64,
                                                            C.POINTER(LowThreshold)]),
    "sono_find_optimal_high_frequency_threshold": (C.c_int, [C.POINTER(CallMatrix), C.c_double, C.c_int64,
                                                             C.POINTER(HighThreshold)]),
    "sono_rms_snr": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_uint32, _f32p, C.c_int64, C.c_int64, C.c_double, C.c_double,
                               _f32p, C.c_size_t, C.c_uint32, _f32p, C.c_double, C.c_double, C.POINTER(SnrResult)]),
    "sono_refine_end_oscillogram": (C.c_int, [_f32p, C.c_size_t, C.c_double, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "sono_frame_calls": (C.c_int, [_segp, C.c_size_t, C.POINTER(C.c_double), C.c_size_t, _f32p, C.c_size_t, C.c_double,
                                   C.c_uint32, C.c_double, C.c_double, C.c_int, C.POINTER(CallFrame), C.c_size_t, _szp]),
    "sono_scan_spectrum_for_frequency": (C.c_int, [_f32p, _f32p, C.c_size_t, C.c_double, C.c_int, C.POINTER(C.c_double),
                                                   C.POINTER(C.c_int64)]),
    "sono_auto_highpass_khz": (C.c_double, [C.c_double]),
    "sono_instant_bandwidth_khz": (C.c_int, [_f32p, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "sono_peak_frequency_khz": (C.c_int, [_f32p, _f32p, C.c_size_t, C.c_int64, C.POINTER(C.c_double)]),
    "sono_highpass_filter": (C.c_int, [_f32p, C.c_size_t, C.c_double, C.c_double, C.c_double, _f32p]),
    "sono_savitzky_golay": (C.c_int, [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double)]),
    "sono_roi_rows": (C.c_int, [_segp, C.c_size_t, C.c_uint32, C.c_double, C.POINTER(C.c_uint64)]),
    "sono_analyze_rois": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, _segp, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int, C.POINTER(RoiTable), C.c_size_t, _f32p]),
    "sono_frame_roi_calls": (C.c_int, [_f32p, _f32p, C.c_size_t, _segp, C.c_size_t, C.POINTER(RoiTable), C.c_double, C.c_uint32,
                                       C.c_double, C.c_double, C.POINTER(RoiCall), C.c_size_t, _szp]),
    "sono_wav_parse": (C.c_int, [_u8p, C.c_size_t, C.POINTER(WavInfo)]),
    "sono_pcm_to_f32_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "sono_wav_decode_channel": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, C.c_uint32, _f32p, C.c_size_t, _szp,
                                          C.POINTER(WavInfo)]),
    "sono_power_spectrum": (C.c_int, [_f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int,
                                      C.c_float, C.c_int, _f32p, C.c_size_t, _szp]),
    "sono_find_peak_frequency": (C.c_float, [_f32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    "sono_wave_peaks": (C.c_int, [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, _szp]),
    "sono_find_global_max": (C.c_float, [_f32p, C.c_size_t]),
    "sono_waveform_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "sono_waveform_destroy": (None, [C.c_void_p]),
    "sono_waveform_resize": (C.c_int, [C.c_void_p, C.c_size_t]),
    "sono_waveform_load_channel": (C.c_int, [C.c_void_p, C.c_size_t, _f32p, C.c_size_t]),
    "sono_waveform_channel_length": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "sono_waveform_num_channels": (C.c_size_t, [C.c_void_p]),
    "sono_waveform_clear": (C.c_int, [C.c_void_p]),
    "sono_waveform_release_memory": (C.c_int, [C.c_void_p]),
    "sono_waveform_peaks_in_range": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                               _f32p, C.c_size_t, _szp]),
}


def _share_torch_hip_runtime():
    """PyTorch wheels carry their own libamdhip64.so.N with the same soname as the
    system ROCm's.  Whichever copy a process loads first serves every later
    user; if libsonostft pulled in the system copy first, a later `import torch`
    would run on a HIP runtime it was not built with and see no GPU.  So when
    PyTorch is installed, its copy is loaded before libsonostft."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded libsonostft.so.  Raises if the in-tree build is missing."""
    global _lib
    if _lib is None:
        # SONO_LIB_PATH: another build of the same library (the host-sanitizer build of build.py --asan)
        path = os.environ.get("SONO_LIB_PATH") or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -m spectrogram_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(status):
    if status != SONO_OK:
        raise SonoError(status, lib().sono_last_error().decode("utf-8", "replace"))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def device_count():
    return lib().sono_device_count()


def frames_for(n_samples, fft_size, noverlap):
    """Frame count rule of lib.rs:118-123."""
    return lib().sono_frames_for(n_samples, fft_size, noverlap)


def set_host_threads(n):
    """Host threads one library call may use (0: the default, min(16, cores / LOCAL_WORLD_SIZE))."""
    _check(lib().sono_set_host_threads(int(n)))


def host_threads():
    return int(lib().sono_host_threads())


class SpectrogramEngine:
    """SpectrogramEngine of spectrogram_wasm.d.ts:4-140, computing on the GPU.

    SpectrogramEngine(fft_size, window_func, alpha=None, device=-1)
    """

    def __init__(self, fft_size, window_func, alpha=None, device=-1):
        h = C.c_void_p()
        _check(lib().sono_engine_create(int(fft_size), str(window_func).encode(), 0 if alpha is None else 1,
                                        0.0 if alpha is None else float(alpha), int(device), C.byref(h)))
        self._h = h

    # -- lifetime ---------------------------------------------------------
    def free(self):
        if getattr(self, "_h", None):
            lib().sono_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def _handle(self):
        if not self._h:
            raise SonoError(SONO_EINVAL, "null pointer passed to rust")   # the wasm glue's use-after-free message
        return self._h

    # -- getters ------------------------------------------------------------
    def get_fft_size(self):
        return lib().sono_engine_fft_size(self._handle())

    def get_freq_bins(self):
        return lib().sono_engine_freq_bins(self._handle())

    def get_num_filters(self):
        return lib().sono_engine_num_filters(self._handle())

    def get_window_values(self):
        out = np.empty(self.get_fft_size(), dtype=np.float32)
        _check(lib().sono_engine_window_values(self._handle(), _p(out), out.size))
        return out

    # -- filter bank ----------------------------------------------------------
    def load_filter_bank(self, flat_weights, num_filters):
        w = _f32(flat_weights)
        _check(lib().sono_engine_load_filter_bank(self._handle(), _p(w), w.size, int(num_filters)))

    def clear_filter_bank(self):
        _check(lib().sono_engine_clear_filter_bank(self._handle()))

    # -- the hot path -------------------------------------------------------------
    def compute_spectrogram(self, audio_data, noverlap):
        """-> Float32Array, frames x fft_size/2 linear magnitudes (lib.rs:113-161)."""
        x = _f32(audio_data)
        n_fft = self.get_fft_size()
        F = frames_for(x.size, n_fft, int(noverlap)) if 0 <= noverlap < n_fft else 0
        out = np.empty(F * (n_fft // 2), dtype=np.float32)
        got = C.c_size_t(0)
        _check(lib().sono_stft_mag_f32(self._handle(), _p(x), x.size, int(noverlap), _p(out), out.size,
                                       C.byref(got)))
        return out[: got.value * (n_fft // 2)]

    def compute_spectrogram_u8(self, audio_data, noverlap, gain_db, range_db):
        """-> Uint8Array, frames x bins dB codes (lib.rs:199-304)."""
        x = _f32(audio_data)
        n_fft = self.get_fft_size()
        nf = self.get_num_filters()
        bins = nf if nf > 0 else n_fft // 2
        F = frames_for(x.size, n_fft, int(noverlap)) if 0 <= noverlap < n_fft else 0
        out = np.empty(F * bins, dtype=np.uint8)
        got = C.c_size_t(0)
        _check(lib().sono_stft_u8(self._handle(), _p(x), x.size, int(noverlap), float(gain_db), float(range_db),
                                  _p(out, _u8p), out.size, C.byref(got)))
        self._last_frames = got.value
        return out[: got.value * bins]

    def get_peaks(self, threshold_ratio):
        cap = max(getattr(self, "_last_frames", 0), 1)
        out = np.empty(cap, dtype=np.uint16)
        got = C.c_size_t(0)
        _check(lib().sono_peaks_u16(self._handle(), float(threshold_ratio), _p(out, _u16p), cap, C.byref(got)))
        return out[: got.value]

    def get_peak_magnitudes(self, threshold_ratio):
        cap = max(getattr(self, "_last_frames", 0), 1)
        out = np.empty(cap, dtype=np.float32)
        got = C.c_size_t(0)
        _check(lib().sono_peak_mags_f32(self._handle(), float(threshold_ratio), _p(out), cap, C.byref(got)))
        return out[: got.value]

    def get_global_max(self):
        return float(lib().sono_global_max(self._handle()))

    # -- image path (cold) ----------------------------------------------------------
    def set_color_map(self, colors):
        c = np.ascontiguousarray(colors, dtype=np.uint8)
        _check(lib().sono_engine_set_color_map(self._handle(), _p(c, _u8p), c.size))

    def set_spectrum_config(self, scale, freq_min, freq_max):
        _check(lib().sono_engine_set_spectrum_config(self._handle(), str(scale).encode(), float(freq_min),
                                                     float(freq_max)))

    def compute_spectrogram_image(self, audio_data, width, height, noverlap, gain_db, range_db):
        x = _f32(audio_data)
        out = np.zeros(int(width) * int(height) * 4, dtype=np.uint8)
        _check(lib().sono_spectrogram_image(self._handle(), _p(x), x.size, int(width), int(height), int(noverlap),
                                            float(gain_db), float(range_db), _p(out, _u8p), out.size))
        return out

    def release_memory(self):
        _check(lib().sono_engine_release_memory(self._handle()))
        self._last_frames = 0

    # -- device-resident forms (batch / server use; no wasm counterpart) -------------
    def stft_u8_dev(self, d_samples, n_streams, stream_stride, samples_per_stream, noverlap, gain_db, range_db,
                    d_out, d_peak_bin=0, d_peak_mag=0, d_gmax_bits=0, stream=0):
        """Launch the fused kernel over samples already in HBM.  Pointer arguments
        are raw device addresses (e.g. torch.Tensor.data_ptr()); asynchronous.
        `stream` is a hipStream_t value (e.g. torch.cuda.Stream().cuda_stream); 0 or
        None selects the engine's own stream, which `synchronize()` waits for."""
        b = Batch(d_samples, n_streams, stream_stride, samples_per_stream)
        _check(lib().sono_stft_u8_dev(self._handle(), C.byref(b), int(noverlap), float(gain_db), float(range_db),
                                      d_out, d_peak_bin or None, d_peak_mag or None, d_gmax_bits or None,
                                      stream or None))

    def stft_u8_pcm16_dev(self, d_pcm, n_streams, stream_stride, samples_per_stream, channels, channel, noverlap,
                          gain_db, range_db, d_out=0, d_peak_bin=0, d_peak_mag=0, d_gmax_bits=0, stream=0):
        """stft_u8_dev on interleaved 16-bit PCM sample frames in HBM (a WAV data chunk as it is): the
        conversion to value / 32768 happens in the kernel's loads.  Strides and lengths count sample frames.
        d_out = 0: features only."""
        _check(lib().sono_stft_u8_pcm16_dev(self._handle(), d_pcm, int(n_streams), int(stream_stride),
                                            int(samples_per_stream), int(channels), int(channel), int(noverlap),
                                            float(gain_db), float(range_db), d_out or None, d_peak_bin or None,
                                            d_peak_mag or None, d_gmax_bits or None, stream or None))

    def stft_band_active_pcm16_dev(self, d_pcm, n_streams, stream_stride, samples_per_stream, channels, channel, noverlap,
                                   min_bin, max_bin, target_energy_sum, d_active, d_band_energy=0, stream=0):
        """stft_band_active_dev (the fused fast scan) on interleaved 16-bit PCM sample frames in HBM."""
        _check(lib().sono_stft_band_active_pcm16_dev(self._handle(), d_pcm, int(n_streams), int(stream_stride),
                                                     int(samples_per_stream), int(channels), int(channel), int(noverlap),
                                                     int(min_bin), int(max_bin), float(target_energy_sum), d_active,
                                                     d_band_energy or None, stream or None))

    def stft_survey_dev(self, d_samples, n_streams, stream_stride, samples_per_stream, min_bin, max_bin, target_energy_sum,
                        d_peak_bin, d_peak_mag, d_gmax_bits, d_active, d_band_energy=0, stream=0):
        """Peak features at hop N/4 and the fast scan's band flags at hop N/2 (one per EVEN frame) in one read of the samples."""
        b = Batch(d_samples, n_streams, stream_stride, samples_per_stream)
        _check(lib().sono_stft_survey_dev(self._handle(), C.byref(b), int(min_bin), int(max_bin), float(target_energy_sum),
                                          d_peak_bin or None, d_peak_mag or None, d_gmax_bits or None, d_active,
                                          d_band_energy or None, stream or None))

    def stft_survey_pcm16_dev(self, d_pcm, n_streams, stream_stride, samples_per_stream, channels, channel, min_bin, max_bin,
                              target_energy_sum, d_peak_bin, d_peak_mag, d_gmax_bits, d_active, d_band_energy=0, stream=0):
        """stft_survey_dev on interleaved 16-bit sample frames."""
        _check(lib().sono_stft_survey_pcm16_dev(self._handle(), d_pcm, int(n_streams), int(stream_stride), int(samples_per_stream),
                                                int(channels), int(channel), int(min_bin), int(max_bin), float(target_energy_sum),
                                                d_peak_bin or None, d_peak_mag or None, d_gmax_bits or None, d_active,
                                                d_band_energy or None, stream or None))

    def stft_features_dev(self, d_samples, n_streams, stream_stride, samples_per_stream, noverlap,
                          d_peak_bin=0, d_peak_mag=0, d_gmax_bits=0, stream=0):
        """stft_u8_dev without the u8 rows: per-frame peak bin / peak magnitude and the global maximum only."""
        b = Batch(d_samples, n_streams, stream_stride, samples_per_stream)
        _check(lib().sono_stft_features_dev(self._handle(), C.byref(b), int(noverlap), d_peak_bin or None,
                                            d_peak_mag or None, d_gmax_bits or None, stream or None))

    def stft_u8_ragged_dev(self, d_samples, offsets, lengths, noverlap, gain_db, range_db, d_out=0, d_peak_bin=0,
                           d_peak_mag=0, d_gmax_bits=0, stream=0):
        """stft_u8_dev over streams of different lengths: stream s = samples [offsets[s], offsets[s] + lengths[s]) of
        d_samples.  Rows are laid out as ragged_rows() says.  d_out = 0: features only."""
        off, ln, b = _ragged(d_samples, offsets, lengths)
        _check(lib().sono_stft_u8_ragged_dev(self._handle(), C.byref(b), int(noverlap), float(gain_db), float(range_db),
                                             d_out or None, d_peak_bin or None, d_peak_mag or None, d_gmax_bits or None,
                                             stream or None))

    def stft_mag_f32_ragged_dev(self, d_samples, offsets, lengths, noverlap, d_out, stream=0):
        off, ln, b = _ragged(d_samples, offsets, lengths)
        _check(lib().sono_stft_mag_f32_ragged_dev(self._handle(), C.byref(b), int(noverlap), d_out, stream or None))

    def stft_band_active_dev(self, d_samples, n_streams, stream_stride, samples_per_stream, noverlap, min_bin, max_bin,
                             target_energy_sum, d_active, d_band_energy=0, stream=0):
        """The fast scan's STFT + band-energy test in one launch (no magnitude matrix)."""
        b = Batch(d_samples, n_streams, stream_stride, samples_per_stream)
        _check(lib().sono_stft_band_active_dev(self._handle(), C.byref(b), int(noverlap), int(min_bin), int(max_bin),
                                               float(target_energy_sum), d_active, d_band_energy or None, stream or None))

    def stft_mag_f32_dev(self, d_samples, n_streams, stream_stride, samples_per_stream, noverlap, d_out, stream=0):
        b = Batch(d_samples, n_streams, stream_stride, samples_per_stream)
        _check(lib().sono_stft_mag_f32_dev(self._handle(), C.byref(b), int(noverlap), d_out, stream or None))

    def synchronize(self):
        _check(lib().sono_engine_synchronize(self._handle()))

    def set_option(self, name, value):
        """Engine switches ("force_generic", "force_any_hop", "fixed_tiles", "balance_debug", "no_streaming", "bank_two_kernels", "host_measure", "no_upload_overlap", "no_measure_overlap", "small_upload_pieces"):
        read from the environment once, when the engine is created; changed here afterwards."""
        _check(lib().sono_engine_set_option(self._handle(), str(name).encode(), int(bool(value))))

    def get_option(self, name):
        v = C.c_int(0)
        _check(lib().sono_engine_get_option(self._handle(), str(name).encode(), C.byref(v)))
        return bool(v.value)


def _ragged(d_samples, offsets, lengths):
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    if off.shape != ln.shape or off.ndim != 1:
        raise ValueError("offsets and lengths must be 1-D arrays of the same length")
    u64p = C.POINTER(C.c_uint64)
    return off, ln, RaggedBatch(d_samples, off.size, off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p))


def ragged_rows(lengths, fft_size, noverlap):
    """First output row of every stream of a ragged batch, and the total: n + 1 values (sono_ragged_rows)."""
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    out = np.zeros(ln.size + 1, np.uint64)
    u64p = C.POINTER(C.c_uint64)
    _check(lib().sono_ragged_rows(ln.ctypes.data_as(u64p), ln.size, int(fft_size), int(noverlap), out.ctypes.data_as(u64p)))
    return out


class WaveformEngine:
    """WaveformEngine of spectrogram_wasm.d.ts:142-211 (host only: O(n) max-abs scans)."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib().sono_waveform_create(C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None):
            lib().sono_waveform_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def resize(self, num_channels):
        _check(lib().sono_waveform_resize(self._h, int(num_channels)))

    def load_channel(self, channel_idx, data):
        x = _f32(data)
        _check(lib().sono_waveform_load_channel(self._h, int(channel_idx), _p(x), x.size))

    def get_peaks_in_range(self, channel_idx, start_sample, end_sample, target_width):
        out = np.empty(max(int(target_width), 1), dtype=np.float32)
        got = C.c_size_t(0)
        _check(lib().sono_waveform_peaks_in_range(self._h, int(channel_idx), int(start_sample), int(end_sample),
                                                  int(target_width), _p(out), out.size, C.byref(got)))
        return out[: got.value]

    def get_channel_length(self, channel_idx):
        return lib().sono_waveform_channel_length(self._h, int(channel_idx))

    def get_num_channels(self):
        return lib().sono_waveform_num_channels(self._h)

    def clear(self):
        _check(lib().sono_waveform_clear(self._h))

    def release_memory(self):
        _check(lib().sono_waveform_release_memory(self._h))


def compute_power_spectrum(audio_data, sample_rate, fft_size, window_type, overlap_percent=None, device=-1):
    """spectrogram_wasm.d.ts:226 / lib.rs:1019-1118 -> Float32Array of fft_size/2+1 dB values (or empty)."""
    x = _f32(audio_data)
    out = np.empty(int(fft_size) // 2 + 1, dtype=np.float32)
    got = C.c_size_t(0)
    _check(lib().sono_power_spectrum(_p(x), x.size, int(sample_rate), int(fft_size), str(window_type).encode(),
                                     0 if overlap_percent is None else 1,
                                     0.0 if overlap_percent is None else float(overlap_percent), int(device),
                                     _p(out), out.size, C.byref(got)))
    return out[: got.value]


def find_peak_frequency_from_spectrum(spectrum, sample_rate, fft_size, flow_hz, fhigh_hz):
    """spectrogram_wasm.d.ts:271 / lib.rs:1132-1179."""
    s = _f32(spectrum)
    return float(lib().sono_find_peak_frequency(_p(s), s.size, int(sample_rate), int(fft_size), float(flow_hz),
                                                float(fhigh_hz)))


def compute_wave_peaks(channel_data, num_peaks):
    """spectrogram_wasm.d.ts:245 / lib.rs:794-824."""
    x = _f32(channel_data)
    out = np.empty(max(int(num_peaks), 1), dtype=np.float32)
    got = C.c_size_t(0)
    _check(lib().sono_wave_peaks(_p(x), x.size, int(num_peaks), _p(out), out.size, C.byref(got)))
    return out[: got.value]


def find_global_max(channel_data):
    """spectrogram_wasm.d.ts:256 / lib.rs:834-840."""
    x = _f32(channel_data)
    return float(lib().sono_find_global_max(_p(x), x.size))


def wav_info(data):
    """RIFF chunk walk of modules/fileLoader.js:7-63 over the bytes of a .wav file -> dict (host only)."""
    b = np.frombuffer(data, dtype=np.uint8)
    info = WavInfo()
    _check(lib().sono_wav_parse(_p(b, _u8p), b.size, C.byref(info)))
    return {k: int(getattr(info, k)) for k, _ in WavInfo._fields_}


def decode_wav(engine, data, channel=0):
    """One channel of a .wav file as float32 (what AudioBuffer.getChannelData gives the application):
    the data chunk is uploaded and converted on the engine's GPU.  -> (samples, info dict)."""
    b = np.frombuffer(data, dtype=np.uint8)
    meta = wav_info(data)
    out = np.empty(max(meta["frames"], 1), dtype=np.float32)
    got = C.c_size_t(0)
    _check(lib().sono_wav_decode_channel(engine._handle(), _p(b, _u8p), b.size, int(channel), _p(out), out.size,
                                         C.byref(got), None))
    return out[: got.value], meta
