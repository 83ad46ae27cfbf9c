"""The CPU oracle against the golden vectors captured from the reference wasm.

This is what pins the oracle: every window table bit-exact, every STFT output
inside the project tolerance, every rule about empty results and frame counts
exact.  Runs on CPU only.
"""
import numpy as np
import pytest

from oracle import sono_oracle as so
from tests import parity


def test_windows_bit_exact(golden):
    g, meta = golden
    ws = parity.cases(meta, "window")
    assert len(ws) >= 40
    for c in ws:
        w = so.create_window(c["window"], c["fft"], 0.16 if c["alpha"] is None else c["alpha"])
        ref = g[c["name"] + "/window"]
        assert np.array_equal(w.view(np.uint32), ref.view(np.uint32)), c["name"]


def test_stft_cases(golden):
    g, meta = golden
    nd = nt = 0
    for c in parity.cases(meta, "stft"):
        e = so.OracleEngine(c["fft"], c["window"], c["alpha"])
        a, b = parity.run_stft_case(e, g, c)
        nd += a
        nt += b
    assert nt > 100000
    assert nd <= parity.U8_MAX_FRACTION * nt, f"{nd} of {nt} u8 bins differ"


def test_frame_count_rule(golden):
    g, meta = golden
    # n < N -> 0 frames; n = N -> 1; n = N+H-1 -> 1; n = N+H -> 2 (lib.rs:118-123)
    assert [so.num_frames(n, 256, 128) for n in (255, 256, 383, 384)] == [0, 1, 1, 2]
    for n in (255, 256, 383, 384):
        assert len(g[f"edge256_n{n}/mag"]) == so.num_frames(n, 256, 128) * 128
    assert so.num_frames(1382400000, 1024, 768) == 5399997
    assert so.num_frames(2500000, 1024, 768) == 9762
    assert so.num_frames(230400000, 4096, 3891) == 1123883


def test_silence_gives_empty_peaks(golden):
    g, _ = golden
    e = so.OracleEngine(1024, "hann")
    assert len(e.get_peaks(0.4)) == 0           # before any u8 call (lib.rs:350-352)
    u8 = e.compute_spectrogram_u8(g["silence1024/x"], 768, 20.0, 80.0)
    assert not u8.any()
    assert e.get_global_max() == 0.0
    assert len(e.get_peaks(0.4)) == 0 and len(e.get_peak_magnitudes(0.0)) == 0
    assert len(g["silence1024/peaks_r0"]) == 0


def test_u8_truncates_not_rounds():
    # (db - lo) * 255/range is truncated (lib.rs:289): an impulse through a
    # rectangular window has |X[k]| = a exactly, so the codes are predictable.
    e = so.OracleEngine(256, "rectangular")
    for target in (191.25, 127.5, 63.75, 0.999, 254.999):
        db = target * 80.0 / 255.0 - 100.0
        a = 10 ** (db / 20.0) * 128.0
        x = np.zeros(256, np.float32)
        x[0] = a
        u8 = e.compute_spectrogram_u8(x, 0, 20.0, 80.0)
        assert abs(int(u8[5]) - int(target)) <= 1 and u8[5] <= int(target) + (target % 1 > 0.99)


def test_power_spectrum(golden):
    g, meta = golden
    for c in parity.cases(meta, "power"):
        n = c["name"]
        s = so.compute_power_spectrum(g[n + "/x"], c["sr"], c["fft"], c["window"], c["overlap"])
        parity.check_power(s, g[n + "/spectrum"], n)
        ref = g[n + "/spectrum"]
        for q, want in zip(c["queries"], g[n + "/peakfreq"]):
            got = so.find_peak_frequency_from_spectrum(ref, c["sr"], c["fft"], q[0], q[1])
            assert np.float32(got) == want, (n, q)


def test_power_spectrum_hop_rule():
    L = so.lib()
    assert L.so_power_spectrum_hop(1024, 0, 0.0) == 256      # None -> 75 % (lib.rs:1032-1034)
    assert L.so_power_spectrum_hop(1024, 1, 0.0) == 256
    assert L.so_power_spectrum_hop(1024, 1, 100.0) == 256
    assert L.so_power_spectrum_hop(1024, 1, 50.0) == 512
    assert L.so_power_spectrum_hop(1024, 1, 99.0) == 10      # trunc(1024 * (1 - 0.99)) in f32
    assert L.so_power_spectrum_hop(256, 1, 99.9) == 1        # at least 1 (lib.rs:1038)


def test_wave_helpers(golden):
    g, meta = golden
    for c in parity.cases(meta, "wave"):
        n = c["name"]
        assert np.array_equal(so.compute_wave_peaks(g[n + "/x"], c["num_peaks"]), g[n + "/peaks"])
        assert np.float32(so.find_global_max(g[n + "/x"])) == g[n + "/gmax"][0]
    assert len(so.compute_wave_peaks(np.zeros(0, np.float32), 5)) == 0
    assert len(so.compute_wave_peaks(np.ones(5, np.float32), 0)) == 0


def test_fft_against_float64():
    rng = np.random.default_rng(3)
    for n in (256, 512, 1024, 2048, 4096):
        x = rng.standard_normal(n).astype(np.float32)
        y = rng.standard_normal(n).astype(np.float32)
        re, im = so.fft_forward(x, y)
        ref = np.fft.fft(x.astype(np.float64) + 1j * y.astype(np.float64))
        err = np.abs((re + 1j * im) - ref).max() / np.abs(ref).max()
        assert err < 2e-6, (n, err)


def test_msun_log10_port():
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(np.log(1e-12), np.log(10.0), 200000)).astype(np.float32)
    out = np.empty_like(x)
    so.lib().so_probe_log10f(x.ctypes.data_as(so._f32p), len(x), out.ctypes.data_as(so._f32p))
    ref = np.log10(x.astype(np.float64)).astype(np.float32)
    ulp = np.abs(out.view(np.int32).astype(np.int64) - ref.view(np.int32))
    assert ulp.max() <= 1
