"""ctypes front end of the CPU restatement (oracle/sono_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; the product package never imports it.

The classes mirror the reference's wasm export surface
(modules/spectrogram_wasm.d.ts:4-271) closely enough that a parity test reads
the same against this oracle and against the HIP-backed engine.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsono_oracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)


def build(force=False):
    """Compile libsono_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "sono_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and (not os.path.exists(src) or os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "libsono_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.so_create_window.argtypes = [C.c_char_p, C.c_size_t, C.c_float, _f32p]
        L.so_create_window.restype = None
        L.so_num_frames.argtypes = [C.c_size_t] * 3
        L.so_num_frames.restype = C.c_size_t
        L.so_plan_create.argtypes = [C.c_size_t]
        L.so_plan_create.restype = C.c_void_p
        L.so_plan_destroy.argtypes = [C.c_void_p]
        L.so_plan_destroy.restype = None
        L.so_fft_forward.argtypes = [C.c_void_p, _f32p, _f32p]
        L.so_fft_forward.restype = None
        L.so_compute_spectrogram.argtypes = [C.c_void_p, _f32p, _f32p, C.c_size_t, C.c_size_t, _f32p]
        L.so_compute_spectrogram.restype = C.c_size_t
        L.so_compute_spectrogram_u8.argtypes = [
            C.c_void_p, _f32p, _f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_float,
            _f32p, C.c_size_t, C.c_size_t, _u8p, _f32p, _f32p]
        L.so_compute_spectrogram_u8.restype = C.c_size_t
        L.so_get_peaks.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_float, _u16p]
        L.so_get_peaks.restype = C.c_size_t
        L.so_get_peak_magnitudes.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_float, _f32p]
        L.so_get_peak_magnitudes.restype = C.c_size_t
        L.so_power_spectrum_hop.argtypes = [C.c_size_t, C.c_int, C.c_float]
        L.so_power_spectrum_hop.restype = C.c_size_t
        L.so_compute_power_spectrum.argtypes = [
            _f32p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_char_p, C.c_int, C.c_float, _f32p]
        L.so_compute_power_spectrum.restype = C.c_size_t
        L.so_find_peak_frequency.argtypes = [_f32p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_float, C.c_float]
        L.so_find_peak_frequency.restype = C.c_float
        L.so_compute_wave_peaks.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p]
        L.so_compute_wave_peaks.restype = C.c_size_t
        L.so_find_global_max.argtypes = [_f32p, C.c_size_t]
        L.so_find_global_max.restype = C.c_float
        L.so_wave_peaks_in_range.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _f32p]
        L.so_wave_peaks_in_range.restype = C.c_size_t
        L.so_probe_log10f.argtypes = [_f32p, C.c_size_t, _f32p]
        L.so_probe_log10f.restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=_f32p):
    return a.ctypes.data_as(t)


def create_window(name, size, alpha=0.16):
    """lib.rs:698-776."""
    w = np.empty(size, dtype=np.float32)
    lib().so_create_window(name.encode(), size, alpha, _ptr(w))
    return w


def num_frames(n, fft_size, noverlap):
    return lib().so_num_frames(n, fft_size, noverlap)


def fft_forward(re, im):
    re = _f32(re).copy()
    im = _f32(im).copy()
    p = lib().so_plan_create(len(re))
    lib().so_fft_forward(p, _ptr(re), _ptr(im))
    lib().so_plan_destroy(p)
    return re, im


class OracleEngine:
    """CPU stand-in for SpectrogramEngine (lib.rs:8-694), hot-path methods only."""

    def __init__(self, fft_size, window_func="hann", alpha=None):
        if fft_size < 2 or fft_size & (fft_size - 1):
            raise ValueError("fft_size must be a power of two")
        self.fft_size = fft_size
        self.alpha = 0.16 if alpha is None else float(alpha)   # lib.rs:47
        self.window = create_window(window_func, fft_size, self.alpha)
        self._plan = lib().so_plan_create(fft_size)
        self._bank = None
        self._num_filters = 0
        self._mags = None
        self._frames = 0
        self._gmax = 0.0

    def __del__(self):
        try:
            if self._plan:
                lib().so_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def get_fft_size(self):
        return self.fft_size

    def get_freq_bins(self):
        return self.fft_size // 2

    def get_num_filters(self):
        return self._num_filters

    def get_window_values(self):
        return self.window.copy()

    def load_filter_bank(self, flat_weights, num_filters):
        self._bank = _f32(flat_weights).copy()
        self._num_filters = int(num_filters)

    def clear_filter_bank(self):
        self._bank = None
        self._num_filters = 0

    def compute_spectrogram(self, audio, noverlap):
        x = _f32(audio)
        if not 0 <= noverlap < self.fft_size:
            raise ValueError("noverlap must be in [0, fft_size)")
        F = num_frames(len(x), self.fft_size, noverlap)
        out = np.empty(F * (self.fft_size // 2), dtype=np.float32)
        lib().so_compute_spectrogram(self._plan, _ptr(self.window), _ptr(x), len(x), noverlap, _ptr(out))
        return out

    def compute_spectrogram_u8(self, audio, noverlap, gain_db, range_db):
        x = _f32(audio)
        if not 0 <= noverlap < self.fft_size:
            raise ValueError("noverlap must be in [0, fft_size)")
        B = self.fft_size // 2
        F = num_frames(len(x), self.fft_size, noverlap)
        use_bank = self._bank is not None and self._num_filters > 0
        out_bins = self._num_filters if use_bank else B
        out = np.empty(F * out_bins, dtype=np.uint8)
        mags = np.empty(F * B, dtype=np.float32)
        gmax = C.c_float(0.0)
        lib().so_compute_spectrogram_u8(
            self._plan, _ptr(self.window), _ptr(x), len(x), noverlap, gain_db, range_db,
            _ptr(self._bank) if use_bank else None, len(self._bank) if use_bank else 0,
            self._num_filters if use_bank else 0,
            _ptr(out, _u8p), _ptr(mags), C.byref(gmax))
        self._mags, self._frames, self._gmax = mags, F, gmax.value
        return out

    def get_peaks(self, threshold_ratio):
        if self._mags is None or self._mags.size == 0:
            return np.empty(0, dtype=np.uint16)
        out = np.empty(self._frames, dtype=np.uint16)
        n = lib().so_get_peaks(_ptr(self._mags), self._frames, self.fft_size // 2, self._gmax,
                               threshold_ratio, _ptr(out, _u16p))
        return out[:n]

    def get_peak_magnitudes(self, threshold_ratio):
        if self._mags is None or self._mags.size == 0:
            return np.empty(0, dtype=np.float32)
        out = np.empty(self._frames, dtype=np.float32)
        n = lib().so_get_peak_magnitudes(_ptr(self._mags), self._frames, self.fft_size // 2, self._gmax,
                                         threshold_ratio, _ptr(out))
        return out[:n]

    def get_global_max(self):
        return float(self._gmax)

    def last_magnitudes(self):
        """Linear magnitudes kept by the last u8 call (lib.rs:299)."""
        return self._mags

    def release_memory(self):
        self._bank = None
        self._num_filters = 0
        self._mags = None
        self._frames = 0
        self._gmax = 0.0


def compute_power_spectrum(audio, sample_rate, fft_size, window_type, overlap_percent=None):
    """lib.rs:1019-1118."""
    x = _f32(audio)
    out = np.empty(fft_size // 2 + 1, dtype=np.float32)
    n = lib().so_compute_power_spectrum(
        _ptr(x), len(x), sample_rate, fft_size, window_type.encode(),
        0 if overlap_percent is None else 1, 0.0 if overlap_percent is None else overlap_percent, _ptr(out))
    return out[:n]


def find_peak_frequency_from_spectrum(spectrum, sample_rate, fft_size, flow_hz, fhigh_hz):
    """lib.rs:1132-1179."""
    s = _f32(spectrum)
    return float(lib().so_find_peak_frequency(_ptr(s), len(s), sample_rate, fft_size, flow_hz, fhigh_hz))


def compute_wave_peaks(channel, num_peaks):
    """lib.rs:794-824."""
    x = _f32(channel)
    out = np.empty(max(num_peaks, 1), dtype=np.float32)
    n = lib().so_compute_wave_peaks(_ptr(x), len(x), num_peaks, _ptr(out))
    return out[:n]


def find_global_max(channel):
    """lib.rs:834-840."""
    x = _f32(channel)
    return float(lib().so_find_global_max(_ptr(x), len(x)))


def wave_peaks_in_range(channel, start_sample, end_sample, target_width):
    """lib.rs:904-955 for one loaded channel."""
    x = _f32(channel)
    out = np.empty(max(target_width, 1), dtype=np.float32)
    n = lib().so_wave_peaks_in_range(_ptr(x), len(x), start_sample, end_sample, target_width, _ptr(out))
    return out[:n]
