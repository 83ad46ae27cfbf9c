"""CPU-side checks of the C ABI: the library loads, exports every declared
symbol, and its host-only entry points (waveform helpers, peak-frequency search,
frame-count rule) agree with the golden vectors.  No GPU compute is attempted."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import spectrogram_amd as sa
from tests import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sonostft.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sono_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 35
    L = C.CDLL(sa.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the Python binding table covers the same set
    assert sorted(sa._SIGNATURES) == names


def test_abi_version_and_frames_rule():
    assert sa.lib().sono_abi_version() == 1
    assert [sa.frames_for(n, 256, 128) for n in (255, 256, 383, 384)] == [0, 1, 1, 2]
    assert sa.frames_for(1382400000, 1024, 768) == 5399997
    assert sa.frames_for(2500000, 1024, 768) == 9762
    assert sa.frames_for(230400000, 4096, 3891) == 1123883
    assert sa.frames_for(5000, 1024, 1024) == 0       # noverlap == N: reference panics; the ABI says 0 frames


def test_no_gpu_means_error_not_fallback():
    if sa.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sa.SonoError) as ei:
        sa.SpectrogramEngine(1024, "hann")
    assert ei.value.status == sa.SONO_EDEVICE
    with pytest.raises(sa.SonoError) as ei:
        sa.compute_power_spectrum(np.zeros(4096, np.float32), 384000, 1024, "hann", None)
    assert ei.value.status == sa.SONO_EDEVICE


def test_bad_fft_size_is_einval():
    for n in (0, 100, 128, 8192, 1000):
        with pytest.raises(sa.SonoError) as ei:
            sa.SpectrogramEngine(n, "hann")
        assert ei.value.status == sa.SONO_EINVAL


def test_wave_helpers_match_reference(golden):
    g, meta = golden
    for c in parity.cases(meta, "wave"):
        n = c["name"]
        assert np.array_equal(sa.compute_wave_peaks(g[n + "/x"], c["num_peaks"]), g[n + "/peaks"])
        assert np.float32(sa.find_global_max(g[n + "/x"])) == g[n + "/gmax"][0]
    assert len(sa.compute_wave_peaks(np.zeros(0, np.float32), 5)) == 0
    assert len(sa.compute_wave_peaks(np.ones(5, np.float32), 0)) == 0


def test_peak_frequency_matches_reference(golden):
    g, meta = golden
    for c in parity.cases(meta, "power"):
        ref = g[c["name"] + "/spectrum"]
        for q, want in zip(c["queries"], g[c["name"] + "/peakfreq"]):
            got = sa.find_peak_frequency_from_spectrum(ref, c["sr"], c["fft"], q[0], q[1])
            assert np.float32(got) == want, (c["name"], q)


def test_waveform_engine_against_oracle(golden):
    from oracle import sono_oracle as so
    g, _ = golden
    x = g["wave_100/x"]
    w = sa.WaveformEngine()
    assert w.get_num_channels() == 0
    w.resize(2)
    w.load_channel(0, x)
    w.load_channel(5, x)                     # out of range: ignored (lib.rs:880-882)
    assert w.get_num_channels() == 2 and w.get_channel_length(0) == len(x) and w.get_channel_length(1) == 0
    for (s, e, tw) in ((0, len(x), 100), (100, 4000, 333), (0, 10 ** 9, 7), (4000, 100, 5), (0, len(x), 0),
                       (0, len(x), 6000)):
        assert np.array_equal(w.get_peaks_in_range(0, s, e, tw), so.wave_peaks_in_range(x, s, e, tw)), (s, e, tw)
    assert np.array_equal(w.get_peaks_in_range(3, 0, 10, 4), np.zeros(4, np.float32))
    w.clear()
    assert w.get_num_channels() == 0
    w.free()
