"""Shared parity checks: one set of rules for the CPU oracle and the HIP path.

Tolerances (the ones the project states for this path):
  * linear magnitudes: |a-b| <= 1e-4*|b| + 1e-6*frame_max  -- 1e-4 relative
    (0.001 dB) wherever a bin is within 40 dB of its frame's peak, loosening to
    0.01 dB at 80 dB below the peak (the floor the u8 law can still see) and to
    an absolute 1e-6 of the frame peak below that, where two different f32 FFTs
    disagree by their own rounding (~1e-7 of the frame peak);
  * u8 frames: never more than 1 LSB apart, and at most 0.1 % of bins differ.  The 0.1 % counts EVERY bin whenever
    the range is 80 dB or less (the application's default and what every golden vector uses).  A wider range (the
    random-configuration test draws 100 dB) lets the code see bins more than 80 dB below their frame's peak, where
    two f32 FFTs differ by their own rounding (1e-7 of the frame's peak, i.e. by several per cent of such a bin) and
    no implementation other than the reference's own FFT can reproduce the code; there, and only there, the 0.1 %
    is counted over the bins within 80 dB of their frame's peak (u8_count_for_bound), while the 1-LSB limit still
    holds for every bin;
  * peak bins, frame counts, empty-result rules: exact.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz")

MAG_RTOL = 1e-4
MAG_FRAME_ATOL = 1e-6
U8_MAX_LSB = 1
U8_MAX_FRACTION = 1e-3

# Flat-spectrum inputs (a single impulse): every bin of a frame has the same
# magnitude up to rounding, so the arg-max is decided by FFT rounding noise and
# is not comparable between two FFT implementations.
ARGMAX_ILL_CONDITIONED = {"impulse1024"}


def load_golden():
    g = np.load(GOLDEN)
    meta = json.loads(bytes(g["__meta__"]).decode())
    return g, meta


def cases(meta, kind):
    return [c for c in meta if c["kind"] == kind]


def check_mag(got, ref, bins, what=""):
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return 0.0
    g2 = got.reshape(-1, bins).astype(np.float64)
    r2 = ref.reshape(-1, bins).astype(np.float64)
    fmax = np.abs(r2).max(axis=1, keepdims=True)
    err = np.abs(g2 - r2)
    tol = MAG_RTOL * np.abs(r2) + MAG_FRAME_ATOL * fmax
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} of {bad.size} magnitudes out of tolerance; "
                           f"worst err/frame_max {np.max(err / np.maximum(fmax, 1e-300)):.3e}")
    return float(np.max(err / np.maximum(fmax, 1e-300)))


def u8_diff(got, ref, what=""):
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return 0, 0
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= U8_MAX_LSB, f"{what}: u8 frames differ by {d.max()} LSB"
    return int((d > 0).sum()), int(d.size)


def u8_count_for_bound(got, ref, ref_mag, bins, range_db):
    """(differing codes, codes) that the 0.1 % bound is applied to: all of them for ranges up to 80 dB; for wider
    ranges the bins within 80 dB of their frame's peak (see the module docstring)."""
    if range_db <= 80.0:
        return int((got != ref).sum()), int(ref.size)
    rows = ref_mag.reshape(-1, bins)
    live = rows >= rows.max(axis=1, keepdims=True) * 1e-4
    return int(((got.reshape(rows.shape) != ref.reshape(rows.shape)) & live).sum()), int(live.sum())


def run_stft_case(engine, g, c):
    """Feed one golden STFT case through `engine` (oracle or HIP engine) and
    check every output against the reference's.  Returns (n_u8_diff, n_u8)."""
    n = c["name"]
    x = g[n + "/x"]
    B = c["fft"] // 2
    if c["num_filters"]:
        engine.load_filter_bank(g[n + "/bank"], c["num_filters"])
    mag = np.asarray(engine.compute_spectrogram(x, c["noverlap"]))
    check_mag(mag, g[n + "/mag"], B, n + "/mag")
    u8 = np.asarray(engine.compute_spectrogram_u8(x, c["noverlap"], c["gain"], c["range"]))
    nd, nt = u8_diff(u8, g[n + "/u8"], n + "/u8")
    gmax_ref = float(g[n + "/gmax"][0])
    gmax = float(engine.get_global_max())
    assert abs(gmax - gmax_ref) <= 1e-6 * abs(gmax_ref), f"{n}: gmax {gmax} vs {gmax_ref}"
    for i, r in enumerate(c["ratios"]):
        pk = np.asarray(engine.get_peaks(r))
        pm = np.asarray(engine.get_peak_magnitudes(r))
        rpk, rpm = g[f"{n}/peaks_r{i}"], g[f"{n}/peakmags_r{i}"]
        assert pk.shape == rpk.shape and pm.shape == rpm.shape, f"{n}: peak array lengths"
        assert pk.dtype == np.uint16 and pm.dtype == np.float32
        if n not in ARGMAX_ILL_CONDITIONED:
            assert np.array_equal(pk, rpk), f"{n}: peak bins differ at ratio {r}"
        else:
            # Hundreds of bins tie for the maximum in the reference's own magnitudes here, and it reports the first of
            # them (lib.rs:370-378).  Which of the tied bins comes out on top in ANOTHER FFT is rounding noise, so the
            # rule is checked as far as it can be: frames in which every bin ties exactly (silence: all magnitudes 0)
            # must give the first bin like the reference, the kept / dropped pattern (0xFFFF) must be the reference's,
            # and any other reported bin must be one of the reference's near-maximal bins -- and the first maximum of
            # this engine's own magnitude row.
            rmag, own = g[n + "/mag"].reshape(-1, B), mag.reshape(-1, B)
            assert np.array_equal(pk == 0xFFFF, rpk == 0xFFFF), f"{n}: kept / dropped frames differ at ratio {r}"
            for f in np.flatnonzero(pk != 0xFFFF):
                if rmag[f].max() == rmag[f].min():
                    assert pk[f] == rpk[f] == 0, f"{n}: frame {f}: all bins tie, the first one is the peak"
                    continue
                assert rmag[f, pk[f]] >= rmag[f].max() * (1 - MAG_RTOL), f"{n}: frame {f}: bin {pk[f]} is not a near-maximal bin of the reference"
                first = int(np.argmax(own[f]))
                assert pk[f] == first or own[f, pk[f]] >= own[f, first] * (1 - 1e-6), f"{n}: frame {f}: bin {pk[f]} vs own first maximum {first}"
        if rpm.size:
            assert np.all(np.abs(pm - rpm) <= 1e-6 * max(gmax_ref, 1e-30) + 1e-6 * np.abs(rpm)), \
                f"{n}: peak magnitudes differ at ratio {r}"
    return nd, nt


def check_power(spec, ref, what=""):
    assert spec.shape == ref.shape, f"{what}: length {spec.shape} vs {ref.shape}"
    if ref.size == 0:
        return
    live = ref > ref.max() - 90.0          # deeper bins are rounding noise of the mean-removed frame
    d = np.abs(spec.astype(np.float64) - ref.astype(np.float64))
    assert d[live].max() <= 0.01, f"{what}: {d[live].max():.4f} dB off within 90 dB of the peak"
