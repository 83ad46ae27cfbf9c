"""spectrogram_amd.power_spectrum: the reference's powerSpectrum.js wrappers (modules/powerSpectrum.js:15-106).
The argument rules are host logic (CPU tests); the calls through the library are GPU tests against the C oracle."""
import numpy as np
import pytest

from oracle import sono_oracle as so
from spectrogram_amd import power_spectrum as ps
from tests import parity


def test_overlap_rule():
    # powerSpectrum.js:34-45
    for auto in ("auto", "", None):
        assert ps._overlap_percent(auto) == 75
    assert ps._overlap_percent(50) == 50 and ps._overlap_percent("50") == 50 and ps._overlap_percent(" 93.7%") == 93
    assert ps._overlap_percent(0) == 0 and ps._overlap_percent(99) == 99 and ps._overlap_percent(12.9) == 12
    assert ps._overlap_percent(-0.5) == 0                       # parseInt("-0.5") is -0, and -0 >= 0
    for out_of_range in (100, -1, "150", "abc", True, [], float("nan"), float("inf")):
        assert ps._overlap_percent(out_of_range) == 75
    assert ps._overlap_percent(1e-7) == 1 and ps._overlap_percent(1e31) == 1    # String(1e-7) = "1e-7": parseInt stops at the e
    assert ps.findOptimalOverlap(None, 384000, 1024, "hann") == 75


def test_empty_inputs_need_no_library():
    assert ps.calculatePowerSpectrumWithOverlap(None, 384000, 1024, "hann") is None
    assert ps.calculatePowerSpectrumWithOverlap(np.zeros(0, np.float32), 384000, 1024, "hann") is None
    assert ps.calculatePowerSpectrum([], 384000, 1024, "hann") is None
    assert ps.findPeakFrequencyFromSpectrum(None, 384000, 1024, 10, 100) is None
    assert ps.findPeakFrequencyFromSpectrum(np.zeros(0, np.float32), 384000, 1024, 10, 100) is None


def tone(n, sr, f_hz, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    return (0.4 * np.sin(2 * np.pi * f_hz * t) + 0.003 * rng.standard_normal(n)).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,percent", [("auto", 75.0), (50, 50.0), ("90", 90.0), (0, 0.0), ("nonsense", 75.0)])
def test_spectrum_against_oracle(overlap, percent):
    sr, N = 384000, 1024
    x = tone(9000, sr, 45300.0, 5)
    spec = ps.calculatePowerSpectrumWithOverlap(x, sr, N, "Hann", overlap)          # the application passes "Hann", "Blackman" ...
    assert spec.dtype == np.float32 and spec.shape == (N // 2 + 1,)
    parity.check_power(spec, so.compute_power_spectrum(x, sr, N, "hann", percent), f"overlap {overlap!r}")
    peak = ps.findPeakFrequencyFromSpectrum(spec, sr, N, 10, 120)
    ref = so.find_peak_frequency_from_spectrum(so.compute_power_spectrum(x, sr, N, "hann", percent), sr, N, 10e3, 120e3) / 1000
    assert abs(peak - 45.3) < 0.05 and abs(peak - ref) < 0.002
    assert ps.findPeakFrequencyFromSpectrum(spec, sr, N, 500, 600) is None        # a band beyond the spectrum: no peak (lib.rs:1147-1150)


@pytest.mark.gpu
def test_single_frame_form_and_short_input():
    sr, N = 256000, 512
    x = tone(4000, sr, 30000.0, 6)
    one = ps.calculatePowerSpectrum(x, sr, N, "Blackman")
    parity.check_power(one, so.compute_power_spectrum(x, sr, N, "blackman", 0.0), "overlap 0")
    assert ps.calculatePowerSpectrumWithOverlap(x[: N - 1], sr, N, "hann") is None      # shorter than a frame: empty -> null
