"""The numeric half of the reference's modules/powerSpectrum.js: its wrappers round `compute_power_spectrum` and
`find_peak_frequency_from_spectrum` (the fourth call site of the hot path, powerSpectrum.js:49), with the reference's
argument rules.  The drawing half (SVG) is the application's own business.

Where the reference returns null this returns None; where it logs an error and returns null because the wasm module
is missing, this lets the library's exception through (there is no CPU path to fall back to)."""
import numpy as np

import spectrogram_amd as sa


def findOptimalOverlap(audioData, sampleRate, fftSize, windowType):
    """powerSpectrum.js:15-18: the automatic overlap is 75 %, whatever the input."""
    return 75


def _overlap_percent(overlap):
    # powerSpectrum.js:34-45: 'auto', '', null and undefined mean 75; otherwise parseInt(overlap, 10) if it is a number
    # in 0..99, else 75
    if overlap is None or overlap == "auto" or overlap == "":
        return 75
    parsed = _parse_int(overlap)
    return parsed if parsed is not None and 0 <= parsed <= 99 else 75


def _parse_int(v):
    """JavaScript's parseInt(v, 10) as far as this argument can meet it: numbers truncate toward zero, strings give their
    leading optionally signed decimal digits (after white space), anything else NaN (None)."""
    if isinstance(v, bool):
        return None                      # parseInt(true) is NaN
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if v != v or v in (float("inf"), float("-inf")):
            return None
        if v != 0.0 and (abs(v) < 1e-6 or abs(v) >= 1e21):
            return _parse_int(repr(v))   # String(1e-7) is "1e-7", and parseInt reads the 1
        return int(v)                    # String(12.9) -> "12.9" -> 12; String(-0.5) -> "-0.5" -> -0
    if isinstance(v, str):
        s = v.lstrip()
        i = 1 if s[:1] in "+-" else 0
        j = i
        while j < len(s) and s[j].isdigit() and s[j].isascii():
            j += 1
        return int(s[:j]) if j > i else None
    return None


def calculatePowerSpectrumWithOverlap(audioData, sampleRate, fftSize, windowType, overlap="auto", device=-1):
    """powerSpectrum.js:24-62: None for missing or empty audio and for an empty spectrum (fewer samples than one frame,
    lib.rs:1027-1029), else the Float32Array of fftSize / 2 + 1 dB values of compute_power_spectrum (lib.rs:1019-1118)
    with the window's name in lower case."""
    if audioData is None or len(audioData) == 0:
        return None
    spectrum = sa.compute_power_spectrum(np.asarray(audioData, dtype=np.float32), sampleRate, fftSize, str(windowType).lower(),
                                         _overlap_percent(overlap), device=device)
    return spectrum if spectrum is not None and len(spectrum) > 0 else None


def calculatePowerSpectrum(audioData, sampleRate, fftSize, windowType, device=-1):
    """powerSpectrum.js:68-73: the same with overlap 0 -- which compute_power_spectrum reads as "not in (0, 100)" and
    answers with its default hop of a quarter frame (lib.rs:1031-1038)."""
    if audioData is None or len(audioData) == 0:
        return None
    return calculatePowerSpectrumWithOverlap(audioData, sampleRate, fftSize, windowType, 0, device=device)


def findPeakFrequencyFromSpectrum(spectrum, sampleRate, fftSize, flowKHz, fhighKHz):
    """powerSpectrum.js:79-106: the peak frequency in kHz between flowKHz and fhighKHz (parabolic interpolation on the dB
    values, lib.rs:1132-1179), None when there is no spectrum or no peak."""
    if spectrum is None or len(spectrum) == 0:
        return None
    hz = sa.find_peak_frequency_from_spectrum(np.asarray(spectrum, dtype=np.float32), sampleRate, fftSize, flowKHz * 1000,
                                              fhighKHz * 1000)
    return hz / 1000 if hz > 0 else None
