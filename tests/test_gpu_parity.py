"""Parity of the HIP path (through the C ABI) against the golden vectors from
the reference wasm and against the CPU oracle.  Needs an MI355X: -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import spectrogram_amd as sa
from oracle import sono_oracle as so
from tests import parity

pytestmark = pytest.mark.gpu

PATHS = ["default", "generic", "any_hop"]


@pytest.fixture(params=PATHS)
def kernel_path(request, monkeypatch):
    """Runs a test once through the normal kernel dispatch, once with the generic
    LDS-staged kernel forced and once with the tiled kernel's any-hop form forced
    where the dispatch would pick the shared-load form (SONO_FORCE_GENERIC and
    SONO_FORCE_ANY_HOP are read when an engine is created; every test creates its engines itself)."""
    monkeypatch.delenv("SONO_FORCE_GENERIC", raising=False)
    monkeypatch.delenv("SONO_FORCE_ANY_HOP", raising=False)
    if request.param == "generic":
        monkeypatch.setenv("SONO_FORCE_GENERIC", "1")
    elif request.param == "any_hop":
        monkeypatch.setenv("SONO_FORCE_ANY_HOP", "1")
    return request.param


def signal(n, seed, amp=0.3):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    return (amp * np.sin(0.2 * t + 1e-5 * t * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)


def test_device_present():
    assert sa.device_count() >= 1


def test_windows_bit_exact(golden):
    g, meta = golden
    for c in parity.cases(meta, "window"):
        e = sa.SpectrogramEngine(c["fft"], c["window"], c["alpha"])
        assert np.array_equal(e.get_window_values().view(np.uint32), g[c["name"] + "/window"].view(np.uint32)), c["name"]
        assert e.get_fft_size() == c["fft"] and e.get_freq_bins() == c["fft"] // 2
        e.free()


def test_golden_stft_cases(golden, kernel_path):
    g, meta = golden
    nd = nt = 0
    for c in parity.cases(meta, "stft"):
        e = sa.SpectrogramEngine(c["fft"], c["window"], c["alpha"])
        a, b = parity.run_stft_case(e, g, c)
        e.free()
        nd += a
        nt += b
    assert nt > 100000
    assert nd <= parity.U8_MAX_FRACTION * nt, f"{nd} of {nt} u8 bins differ from the reference ({kernel_path})"


def test_golden_power_spectrum(golden):
    g, meta = golden
    for c in parity.cases(meta, "power"):
        n = c["name"]
        s = sa.compute_power_spectrum(g[n + "/x"], c["sr"], c["fft"], c["window"], c["overlap"])
        parity.check_power(s, g[n + "/spectrum"], n)
        if len(s):
            for q, want in zip(c["queries"], g[n + "/peakfreq"]):
                got = sa.find_peak_frequency_from_spectrum(s, c["sr"], c["fft"], q[0], q[1])
                # the peak frequency is a parabolic fit on dB values that agree to ~1e-4 dB
                assert abs(got - float(want)) <= 2.0, (n, q, got, want)


@pytest.mark.parametrize("N,V", [(256, 0), (256, 255), (512, 256), (512, 137), (1024, 768), (1024, 512),
                                 (1024, 992), (1024, 819), (2048, 1536), (2048, 2047), (4096, 3891), (4096, 0)])
def test_against_oracle(kernel_path, N, V):
    H = N - V
    frames = 37 if H > 4 else 300
    x = signal(N + (frames - 1) * H + min(H - 1, 3), N + V)
    o = so.OracleEngine(N, "hann")
    e = sa.SpectrogramEngine(N, "hann")
    parity.check_mag(e.compute_spectrogram(x, V), o.compute_spectrogram(x, V), N // 2, f"N={N} V={V}")
    u8 = e.compute_spectrogram_u8(x, V, 20.0, 80.0)
    ref = o.compute_spectrogram_u8(x, V, 20.0, 80.0)
    nd, nt = parity.u8_diff(u8, ref, f"N={N} V={V}")
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    assert np.array_equal(e.get_peaks(0.4), o.get_peaks(0.4))
    assert np.array_equal(e.get_peaks(0.0), o.get_peaks(0.0))
    assert np.allclose(e.get_peak_magnitudes(0.0), o.get_peak_magnitudes(0.0), rtol=1e-6, atol=0)
    assert abs(e.get_global_max() - o.get_global_max()) <= 1e-6 * o.get_global_max()
    e.free()


@pytest.mark.parametrize("N,V", [(256, 192), (1024, 768), (2048, 1536), (4096, 3891)])
@pytest.mark.parametrize("window,alpha", [("bartlett", None), ("bartlettHann", None), ("blackman", None), ("blackman", 0.05),
                                          ("cosine", None), ("gauss", None), ("gauss", 0.4), ("hamming", None), ("lanczos", None),
                                          ("rectangular", None), ("triangular", None), ("no such window", None)])
def test_every_window_at_every_size_against_oracle(N, V, window, alpha):
    """The ten windows of lib.rs:698-776 (and the fallback to Hann for an unknown name) at the small, the headline and
    both wide sizes: the window tables are bit-identical to the reference's (test_windows_bit_exact), here the whole
    transform runs with each of them against the C oracle, which the golden vectors pin."""
    H = N - V
    x = signal(N + 21 * H + 5, N + len(window), amp=0.5)
    o = so.OracleEngine(N, window, alpha)
    e = sa.SpectrogramEngine(N, window, alpha)
    assert np.array_equal(e.get_window_values().view(np.uint32), np.asarray(o.get_window_values()).view(np.uint32))
    parity.check_mag(e.compute_spectrogram(x, V), o.compute_spectrogram(x, V), N // 2, f"{window} N={N}")
    nd, nt = parity.u8_diff(e.compute_spectrogram_u8(x, V, 20.0, 80.0), o.compute_spectrogram_u8(x, V, 20.0, 80.0), f"{window} N={N}")
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    assert np.array_equal(e.get_peaks(0.0), o.get_peaks(0.0))
    e.free()


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_against_oracle(seed):
    """Seeded random draws of everything the entry points take -- fft size, overlap (0 ... N-1), length (shorter than a
    frame, exactly one frame, ragged ends), window and alpha, gain and range, signal level (down to 1e-6 and up to 30),
    signal kind -- each against the C oracle with the project's tolerances.  A peak bin may differ from the oracle's only
    where the oracle's own magnitudes at the two bins agree to 1e-6: two FFTs then rank them by rounding noise."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([256, 512, 1024, 2048, 4096]))
    V = int(rng.choice([0, N // 2, N - N // 4, N - 32, int(N * 0.95), int(rng.integers(0, N)), N - 1]))
    H = N - V
    frames = int(rng.integers(0, 40)) if H >= 8 else int(rng.integers(0, 400))
    n = 0 if frames == 0 and rng.random() < 0.3 else max(0, N + (frames - 1) * H + int(rng.integers(0, H))) if frames else int(rng.integers(1, N))
    window = str(rng.choice(["hann", "hamming", "blackman", "bartlett", "bartlettHann", "cosine", "gauss", "lanczos", "rectangular", "triangular"]))
    alpha = None if rng.random() < 0.7 else float(rng.uniform(0.05, 0.5))
    gain, rng_db = float(rng.choice([20.0, 0.0, 35.5, -10.0])), float(rng.choice([80.0, 60.0, 100.0, 45.25]))
    level = float(10.0 ** rng.uniform(-6, 1.5))
    t = np.arange(n)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        x = np.sin(rng.uniform(0.01, 3.0) * t + rng.uniform(0, 1e-4) * t * t) + 0.02 * rng.standard_normal(n)
    elif kind == 1:
        x = rng.standard_normal(n)
    elif kind == 2:
        x = 0.001 * rng.standard_normal(n)
        if n:
            x[rng.integers(0, n, size=max(1, n // 997))] += 1.0          # clicks
    else:
        x = np.sin(0.7 * t) * (t % (3 * H + 1) < H) + 1e-4 * rng.standard_normal(n)      # bursts next to quiet frames
    x = (level * x).astype(np.float32)
    what = f"seed {seed}: N={N} V={V} n={n} {window} alpha={alpha} gain={gain} range={rng_db} level={level:.3g} kind={kind}"
    o = so.OracleEngine(N, window, alpha)
    e = sa.SpectrogramEngine(N, window, alpha)
    ref_mag = np.asarray(o.compute_spectrogram(x, V))
    mag = e.compute_spectrogram(x, V)
    assert mag.shape == ref_mag.shape == (sa.frames_for(n, N, V) * (N // 2),), what
    parity.check_mag(mag, ref_mag, N // 2, what)
    u8 = e.compute_spectrogram_u8(x, V, gain, rng_db)
    ref = np.asarray(o.compute_spectrogram_u8(x, V, gain, rng_db))
    nd, nt = parity.u8_diff(u8, ref, what)                   # never more than 1 LSB, wherever
    if nt:
        nd = parity.u8_count_for_bound(u8, ref, ref_mag, N // 2, rng_db)
        assert nd[0] <= max(2, parity.U8_MAX_FRACTION * nd[1]), what
    gmax = o.get_global_max()
    assert abs(e.get_global_max() - gmax) <= 1e-6 * gmax, what
    for ratio in (0.0, 0.4):
        pk, rpk = e.get_peaks(ratio), np.asarray(o.get_peaks(ratio))
        assert pk.shape == rpk.shape, what
        if pk.size:
            rows = ref_mag.reshape(-1, N // 2)
            keep = rpk != 0xFFFF
            # a frame whose peak sits within 1e-6 of the threshold may be kept by one and dropped by the other
            edge = np.abs(rows.max(axis=1) - np.float32(gmax * ratio)) <= 1e-6 * gmax
            assert np.array_equal((pk != 0xFFFF) | edge, keep | edge), what
            for f in np.flatnonzero((pk != rpk) & keep & (pk != 0xFFFF)):
                assert abs(rows[f, pk[f]] - rows[f, rpk[f]]) <= 1e-6 * rows[f].max(), f"{what}: frame {f}: bins {pk[f]} / {rpk[f]}"
        pm, rpm = e.get_peak_magnitudes(ratio), np.asarray(o.get_peak_magnitudes(ratio))
        assert pm.shape == rpm.shape, what
        if pm.size:
            both = (pm > 0) & (rpm > 0)
            assert np.all(np.abs(pm[both] - rpm[both]) <= 1e-6 * gmax + 1e-6 * rpm[both]), what
    e.free()


def test_empty_inputs_and_errors(kernel_path):
    e = sa.SpectrogramEngine(1024, "hann")
    assert len(e.get_peaks(0.4)) == 0 and len(e.get_peak_magnitudes(0.0)) == 0      # before any u8 call
    assert e.get_global_max() == 0.0
    short = np.ones(1023, np.float32)
    assert len(e.compute_spectrogram(short, 768)) == 0                                # n < N: empty, no error
    assert len(e.compute_spectrogram_u8(short, 768, 20.0, 80.0)) == 0
    assert len(e.compute_spectrogram(np.zeros(0, np.float32), 0)) == 0
    for bad in (1024, 1025, 5000):                                                   # reference panics (lib.rs:118-120)
        with pytest.raises(sa.SonoError) as ei:
            e.compute_spectrogram(np.ones(4096, np.float32), bad)
        assert ei.value.status == sa.SONO_EINVAL
    sil = np.zeros(1024 + 3 * 256, np.float32)
    u8 = e.compute_spectrogram_u8(sil, 768, 20.0, 80.0)
    assert len(u8) == 4 * 512 and not u8.any()
    assert e.get_global_max() == 0.0 and len(e.get_peaks(0.0)) == 0                   # gmax <= 0: empty (lib.rs:350)
    # a later call replaces the state; release_memory clears it
    x = signal(1024 + 5 * 256, 3)
    e.compute_spectrogram_u8(x, 768, 20.0, 80.0)
    assert len(e.get_peaks(0.4)) == 6
    e.release_memory()
    assert len(e.get_peaks(0.4)) == 0 and e.get_global_max() == 0.0
    e.free()
    with pytest.raises(sa.SonoError):
        e.get_fft_size()                                                             # use after free


def test_u8_truncates_not_rounds(kernel_path):
    # an impulse through a rectangular window: |X[k]| = a for every k
    e = sa.SpectrogramEngine(256, "rectangular")
    o = so.OracleEngine(256, "rectangular")
    for target in (191.25, 127.5, 63.75, 10.5, 254.5):
        db = target * 80.0 / 255.0 - 100.0
        x = np.zeros(256, np.float32)
        x[0] = 10 ** (db / 20.0) * 128.0
        u8 = e.compute_spectrogram_u8(x, 0, 20.0, 80.0)
        ref = o.compute_spectrogram_u8(x, 0, 20.0, 80.0)
        assert np.array_equal(u8, ref) and u8[7] == int(target)
    e.free()


def test_clamps_and_extreme_parameters(kernel_path):
    x = signal(1024 + 7 * 256, 9, amp=0.9)
    x[2000:2100] = 0.0
    for gain, rng in ((20.0, 80.0), (0.0, 100.0), (30.0, 60.0), (-10.0, 40.0), (60.0, 250.0)):
        e = sa.SpectrogramEngine(1024, "hann")
        o = so.OracleEngine(1024, "hann")
        nd, nt = parity.u8_diff(e.compute_spectrogram_u8(x, 768, gain, rng), o.compute_spectrogram_u8(x, 768, gain, rng),
                                f"gain={gain} range={rng}")
        assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
        e.free()


def test_filter_bank_quirk(golden):
    g, meta = golden
    c = [c for c in parity.cases(meta, "stft") if c["name"] == "bank256"][0]
    e = sa.SpectrogramEngine(c["fft"], c["window"], c["alpha"])
    parity.run_stft_case(e, g, c)
    assert e.get_num_filters() == 16
    e.clear_filter_bank()
    assert e.get_num_filters() == 0
    assert len(e.compute_spectrogram_u8(g["bank256/x"], c["noverlap"], 20.0, 80.0)) == 4 * 128
    e.free()


def test_filter_bank_on_a_recording_longer_than_the_magnitude_scratch():
    """With a bank loaded the magnitudes are formed piece by piece (2 GiB of them at a time: 262 144 frames at fft 4096).
    A recording of 300 000 frames gives what its two parts give when each is transformed alone: codes, peaks and the
    global maximum."""
    N, hop, nf = 4096, 64, 24
    piece, total = 262144, 300000
    rng = np.random.default_rng(11)
    n = (total - 1) * hop + N
    t = np.arange(n, dtype=np.float64)
    x = (0.4 * np.sin(2 * np.pi * (0.03 + 0.2 * t / n) * t) * (0.2 + 0.8 * (t / n))).astype(np.float32)
    x += rng.normal(0, 0.01, n).astype(np.float32)
    bank = np.zeros((nf, N // 2 + 1), np.float32)
    pos = np.sort(rng.uniform(0, N // 2 - 1, nf))
    for i, q in enumerate(pos):
        o = int(q)
        bank[i, o], bank[i, o + 1] = 1 - (q - o), q - o
    e = sa.SpectrogramEngine(N, "hann")
    e.load_filter_bank(bank.ravel(), nf)
    whole = e.compute_spectrogram_u8(x, N - hop, 20.0, 80.0).copy()
    whole_bins, whole_mags, whole_max = e.get_peaks(0.0).copy(), e.get_peak_magnitudes(0.0).copy(), e.get_global_max()
    assert whole.size == total * nf
    first = e.compute_spectrogram_u8(x[:(piece - 1) * hop + N], N - hop, 20.0, 80.0).copy()
    first_bins, first_mags, first_max = e.get_peaks(0.0).copy(), e.get_peak_magnitudes(0.0).copy(), e.get_global_max()
    second = e.compute_spectrogram_u8(x[piece * hop:], N - hop, 20.0, 80.0).copy()
    second_bins, second_mags, second_max = e.get_peaks(0.0).copy(), e.get_peak_magnitudes(0.0).copy(), e.get_global_max()
    assert first.size == piece * nf and second.size == (total - piece) * nf
    assert np.array_equal(whole, np.concatenate([first, second]))
    assert np.array_equal(whole_bins, np.concatenate([first_bins, second_bins]))
    assert np.array_equal(whole_mags, np.concatenate([first_mags, second_mags]))
    assert whole_max == max(first_max, second_max) and whole.max() > 100
    e.free()


@pytest.mark.parametrize("N,nf,kind", [(1024, 64, "two_tap"), (1024, 512, "two_tap"), (512, 40, "dense"), (256, 7, "short"),
                                        (2048, 33, "sparse_signed"), (4096, 24, "empty_rows"), (4096, 2048, "two_tap"),
                                        (4096, 1500, "sparse_signed")])
def test_filter_bank_shapes_against_the_oracle(N, nf, kind):
    """apply_filter_bank (lib.rs:310-334) on the bank shapes the non-zero tap list has to get right: the application's
    two-tap banks (rows of N/2 + 1 weights used with stride N/2, spectrogram.esm.js:1197-1232), a dense bank, one shorter
    than filters x bins (the reference's bounds check), signed and zero weights, filters without any tap.  The sums are
    the reference's bit for bit, so the codes differ only where the magnitudes themselves do."""
    B = N // 2
    rng = np.random.default_rng(N + nf)
    if kind == "two_tap":
        bank = np.zeros((nf, B + 1), np.float32)
        pos = np.sort(rng.uniform(0, B - 1, nf))
        for i, p in enumerate(pos):
            o = int(p)
            bank[i, o], bank[i, o + 1] = 1 - (p - o), p - o
    elif kind == "dense":
        bank = rng.uniform(0.0, 0.02, (nf, B)).astype(np.float32)
    elif kind == "short":
        bank = rng.uniform(0.0, 0.05, nf * B - B - 17).astype(np.float32)
    elif kind == "sparse_signed":
        bank = np.where(rng.random((nf, B)) < 0.03, rng.uniform(-0.2, 1.0, (nf, B)), 0.0).astype(np.float32)
        bank[3, 5] = -0.0
    else:
        bank = np.zeros((nf, B), np.float32)
        bank[::3, 10:14] = 0.25
    x = signal(N + 45 * (N // 4), N + nf, amp=0.7)
    e = sa.SpectrogramEngine(N, "hann")
    o = so.OracleEngine(N, "hann")
    e.load_filter_bank(bank.ravel(), nf)
    o.load_filter_bank(bank.ravel(), nf)
    got, want = e.compute_spectrogram_u8(x, N - N // 4, 20.0, 80.0), o.compute_spectrogram_u8(x, N - N // 4, 20.0, 80.0)
    assert got.size == want.size == 46 * nf
    nd, nt = parity.u8_diff(got, want, f"{kind} bank")
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    assert np.array_equal(e.get_peaks(0.3), o.get_peaks(0.3))
    # a second bank on the same engine replaces the first one's taps
    e.load_filter_bank(np.full(B, 0.01, np.float32), 1)
    o.load_filter_bank(np.full(B, 0.01, np.float32), 1)
    nd, nt = parity.u8_diff(e.compute_spectrogram_u8(x, N - N // 4, 20.0, 80.0), o.compute_spectrogram_u8(x, N - N // 4, 20.0, 80.0), "one filter")
    assert nt == 46 and nd <= 2
    e.free()


@pytest.mark.parametrize("nf,V,kind", [(512, 768, "two_tap"), (512, 700, "two_tap"), (64, 0, "two_tap"), (300, 768, "three_tap"),
                                       (36, 512, "empty_rows")])
def test_filter_bank_in_the_transforms_epilogue_equals_the_two_kernel_form(nf, V, kind):
    """At fft 1024 a sparse bank (at most 2 048 taps, filters a multiple of four) is applied inside the STFT kernel's
    epilogue, from the power image in LDS; the engine option "bank_two_kernels" keeps the older form (magnitude matrix in
    HBM, then the bank kernel).  Same magnitudes, same taps in the same order, same quantiser: the rows are equal bit
    for bit, at the shared-load hop and at any other, and so are the peak magnitudes and the global maximum."""
    N, B = 1024, 512
    rng = np.random.default_rng(nf + V)
    taps = {"two_tap": 2, "three_tap": 3, "empty_rows": 2}[kind]
    bank = np.zeros((nf, B + 1), np.float32)
    for i, p in enumerate(np.sort(rng.uniform(0, B - taps, nf))):
        if kind == "empty_rows" and i % 3 == 0:
            continue
        o = int(p)
        w = rng.uniform(0.05, 1.0, taps)
        bank[i, o:o + taps] = (w / w.sum()).astype(np.float32)
    H = N - V
    x = signal(N + 203 * H + 5, nf + V, amp=0.7)
    x[N + 20 * H: N + 23 * H] *= 200.0                          # a loud stretch beside quiet frames
    x[: N + H] = 0.0                                            # silent frames
    res = []
    for two in (False, True):
        e = sa.SpectrogramEngine(N, "hann")
        e.set_option("bank_two_kernels", two)
        e.load_filter_bank(bank.ravel(), nf)
        u8 = e.compute_spectrogram_u8(x, V, 20.0, 80.0).copy()
        res.append((u8, e.get_peak_magnitudes(0.0).copy(), np.float32(e.get_global_max()), e.get_peaks(0.0).copy()))
        e.free()
    assert res[0][0].size == 204 * nf and res[0][0].max() > 100
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    # peak bins: the fused form ranks |2X|^2, the bank kernel the rounded magnitudes (first maximum): equal except on a
    # near-tie of two bins, where both name a bin that holds the row's largest magnitude
    assert (res[0][3] != res[1][3]).sum() <= 1
    o = so.OracleEngine(N, "hann")
    o.load_filter_bank(bank.ravel(), nf)
    nd, nt = parity.u8_diff(res[0][0], o.compute_spectrogram_u8(x, V, 20.0, 80.0), f"{kind} bank in the epilogue")
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)


@pytest.mark.parametrize("N,V,width,height", [(256, 192, 40, 30), (1024, 768, 33, 64), (512, 0, 7, 5)])
def test_image_against_the_restatement(N, V, width, height):
    """compute_spectrogram_image (lib.rs:494-616) pixel by pixel against oracle/image_oracle.py on the C oracle's
    magnitudes.  A red channel that equals the colour index shows how far a pixel's index is off: the indices agree
    except where a frame's u8 code sits on a rounding boundary (the two FFTs differ in the last bits), and then by one.
    Parity unpinned: the reference has neither a caller nor a fixture for this export."""
    from oracle import image_oracle as io
    x = signal(N + 57 * (N - V) + 3, N + width, amp=0.6)
    cmap = np.zeros((256, 4), np.uint8)
    cmap[:, 0] = np.arange(256)
    cmap[:, 1] = 255 - np.arange(256)
    cmap[:, 2] = (np.arange(256) * 7) % 256
    cmap[:, 3] = 255
    e = sa.SpectrogramEngine(N, "hann")
    e.set_color_map(cmap.ravel())
    o = so.OracleEngine(N, "hann")
    mags = np.asarray(o.compute_spectrogram(x, V)).reshape(-1, N // 2)
    for gain, rng in ((20.0, 80.0), (0.0, 60.0)):
        got = e.compute_spectrogram_image(x, width, height, V, gain, rng).reshape(height, width, 4)
        want = io.spectrogram_image(mags, cmap.ravel(), width, height, gain, rng).reshape(height, width, 4)
        d = np.abs(got[..., 0].astype(int) - want[..., 0].astype(int))
        assert d.max() <= 1 and (d > 0).mean() <= 0.02, (d.max(), (d > 0).mean())
        same = d == 0
        assert np.array_equal(got[same], want[same]) and (got[..., 3] == 255).all()
        assert want[..., 0].max() > 50                          # the picture is not blank
    e.free()


def test_image_path():
    e = sa.SpectrogramEngine(256, "hann")
    x = signal(256 + 40 * 64, 4)
    img = e.compute_spectrogram_image(x, 32, 24, 192, 20.0, 80.0)
    assert img.shape == (32 * 24 * 4,) and not img.any()          # no colour map: zeros (lib.rs:504-506)
    e.set_color_map(np.zeros(10, np.uint8))                       # wrong size: ignored (lib.rs:452-454)
    assert not e.compute_spectrogram_image(x, 32, 24, 192, 20.0, 80.0).any()
    cmap = np.zeros((256, 4), np.uint8)
    cmap[:, 0] = np.arange(256)
    cmap[:, 3] = 255
    e.set_color_map(cmap.ravel())
    img = e.compute_spectrogram_image(x, 32, 24, 192, 20.0, 80.0).reshape(24, 32, 4)
    assert (img[..., 3] == 255).all() and img[..., 0].max() > 100 and (img[..., 1] == 0).all()
    # the tone (0.2 rad/sample ~ bin 8 of 128) sits near the bottom of the image, not the top
    col = img[:, 5, 0].astype(int)
    assert col[-4:].max() > col[:4].max()
    e.free()


def test_batch_dev_matches_single_streams(kernel_path):
    import torch
    N, V, per, stride, S = 1024, 768, 1024 + 21 * 256 + 17, 8192, 5
    buf = np.zeros(S * stride, np.float32)
    for s in range(S):
        buf[s * stride: s * stride + per] = signal(per, 50 + s)
    dx = torch.from_numpy(buf).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    F = sa.frames_for(per, N, V)
    out = torch.zeros(S * F * 512, dtype=torch.uint8, device="cuda")
    pb = torch.zeros(S * F, dtype=torch.int16, device="cuda")
    pm = torch.zeros(S * F, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    mag = torch.zeros(S * F * 512, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = torch.cuda.Stream()                 # a caller-supplied HIP stream
    e.stft_u8_dev(dx.data_ptr(), S, stride, per, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(),
                  gm.data_ptr(), st.cuda_stream)
    e.stft_mag_f32_dev(dx.data_ptr(), S, stride, per, V, mag.data_ptr(), st.cuda_stream)
    st.synchronize()
    out, mag = out.cpu().numpy().reshape(S, -1), mag.cpu().numpy().reshape(S, -1)
    pb = pb.cpu().numpy().view(np.uint16).reshape(S, F)
    pm = pm.cpu().numpy().reshape(S, F)
    gmax = gm.cpu().numpy().view(np.float32)[0]
    best = 0.0
    for s in range(S):
        xs = buf[s * stride: s * stride + per]
        assert np.array_equal(out[s], e.compute_spectrogram_u8(xs, V, 20.0, 80.0))     # same kernel: bit-exact
        assert np.array_equal(pb[s], e.get_peaks(0.0)) and np.array_equal(pm[s], e.get_peak_magnitudes(0.0))
        best = max(best, e.get_global_max())
        assert np.array_equal(mag[s], e.compute_spectrogram(xs, V))
    assert gmax == np.float32(best)
    e.free()


def test_chunks_with_halo_equal_one_call(kernel_path):
    # time chunks that overlap by N - hop samples give the frames of the unchunked call bit for bit
    N, V = 1024, 768
    H = N - V
    x = signal(N + 199 * H, 77)
    e = sa.SpectrogramEngine(N, "hann")
    whole = e.compute_spectrogram_u8(x, V, 20.0, 80.0).reshape(-1, 512)
    gwhole = e.get_global_max()
    parts, gm = [], 0.0
    for f0 in range(0, 200, 64):
        f1 = min(f0 + 64, 200)
        chunk = x[f0 * H: (f1 - 1) * H + N]
        parts.append(e.compute_spectrogram_u8(chunk, V, 20.0, 80.0).reshape(-1, 512))
        gm = max(gm, e.get_global_max())
    assert np.array_equal(np.concatenate(parts), whole) and gm == gwhole
    e.free()


def test_unaligned_input_pointer(kernel_path):
    # a view that starts 4 bytes into an allocation: frames must not assume 16-byte alignment
    N, V = 1024, 768
    base = signal(N + 30 * 256 + 3, 5)
    o = so.OracleEngine(N, "hann")
    e = sa.SpectrogramEngine(N, "hann")
    for shift in (1, 2, 3):
        x = base[shift:]
        parity.check_mag(e.compute_spectrogram(x, V), o.compute_spectrogram(x, V), 512, f"shift {shift}")
    e.free()


def test_misaligned_output_pointers_are_refused():
    # rows are written with dword (u8) and 4-dword (magnitude) stores: an output that starts off such a boundary
    # is an argument error, not a silent fallback to stores the pointer does not allow
    import torch
    N, V = 1024, 768
    x = torch.from_numpy(signal(N + 15 * 256, 7)).cuda()
    F = sa.frames_for(x.numel(), N, V)
    e = sa.SpectrogramEngine(N, "hann")
    u8 = torch.zeros(F * 512 + 16, dtype=torch.uint8, device="cuda")
    mag = torch.zeros(F * 512 + 16, dtype=torch.float32, device="cuda")
    for off in (1, 2, 3):
        with pytest.raises(sa.SonoError) as ei:
            e.stft_u8_dev(x.data_ptr(), 1, x.numel(), x.numel(), V, 20.0, 80.0, u8.data_ptr() + off, 0, 0, 0)
        assert ei.value.status == 1 and "4-byte aligned" in str(ei.value)
    for off in (4, 8, 12):
        with pytest.raises(sa.SonoError) as ei:
            e.stft_mag_f32_dev(x.data_ptr(), 1, x.numel(), x.numel(), V, mag.data_ptr() + off)
        assert ei.value.status == 1 and "16-byte aligned" in str(ei.value)
    # aligned offsets are fine, and nothing in front of them is touched
    e.stft_u8_dev(x.data_ptr(), 1, x.numel(), x.numel(), V, 20.0, 80.0, u8.data_ptr() + 4, 0, 0, 0)
    e.stft_mag_f32_dev(x.data_ptr(), 1, x.numel(), x.numel(), V, mag.data_ptr() + 16)
    e.synchronize()
    assert u8[:4].sum().item() == 0 and mag[:4].abs().sum().item() == 0 and u8[4:].any().item() and mag[4:].any().item()
    e.free()


def test_time_chunks_of_the_sharding_layer_equal_one_call(kernel_path):
    # one recording cut into 3 chunks the way sharding.frame_chunks cuts it for 3 GPUs: rows bit-identical to
    # the unchunked call, global maximum = max over the chunks (the value reduce_global_max all-reduces)
    from spectrogram_amd import sharding as sh
    N, V = 512, 384
    H = N - V
    x = signal(N + 333 * H + 11, 31)
    e = sa.SpectrogramEngine(N, "hamming")
    whole = e.compute_spectrogram_u8(x, V, 20.0, 80.0).reshape(-1, N // 2)
    gwhole, pk = e.get_global_max(), e.get_peak_magnitudes(0.0)
    parts, gm, pks = [], 0.0, []
    for fr in sh.frame_chunks(whole.shape[0], 3):
        s0, s1 = sh.chunk_samples(fr, N, H)
        parts.append(e.compute_spectrogram_u8(x[s0:s1], V, 20.0, 80.0).reshape(-1, N // 2))
        gm = max(gm, e.get_global_max())
        pks.append(e.get_peak_magnitudes(0.0))
    assert np.array_equal(np.concatenate(parts), whole) and gm == gwhole
    assert np.array_equal(np.concatenate(pks), pk)
    e.free()


TILED = [(256, 192), (512, 256), (1024, 768), (2048, 1536), (4096, 3891)]


@pytest.mark.parametrize("N,V", TILED)
def test_quiet_frames_next_to_loud_ones(N, V):
    # frames are transformed two at a time: a loud frame must not leak rounding noise into its quiet partner
    H = N - V
    rng = np.random.default_rng(3)
    n = N + 45 * H
    x = (1e-5 * rng.standard_normal(n)).astype(np.float32)
    t = np.arange(n)
    burst = (t > 9 * H) & (t < 9 * H + N // 2)
    x[burst] += (0.8 * np.sin(0.7 * t[burst])).astype(np.float32)
    o = so.OracleEngine(N, "hann")
    e = sa.SpectrogramEngine(N, "hann")
    parity.check_mag(e.compute_spectrogram(x, V), o.compute_spectrogram(x, V), N // 2, f"quiet/loud N={N}")
    nd, nt = parity.u8_diff(e.compute_spectrogram_u8(x, V, 20.0, 80.0), o.compute_spectrogram_u8(x, V, 20.0, 80.0))
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    assert np.array_equal(e.get_peaks(0.0), o.get_peaks(0.0))
    e.free()


@pytest.mark.parametrize("N,V", TILED)
def test_all_zero_frames_stay_exactly_zero(N, V):
    H, B = N - V, N // 2
    x = np.zeros(N + 25 * H, np.float32)
    x[N + 2 * H:] = signal(len(x) - (N + 2 * H), 9)          # frames 0..2 see only zeros
    o = so.OracleEngine(N, "hann")
    e = sa.SpectrogramEngine(N, "hann")
    mag = e.compute_spectrogram(x, V)
    assert not mag[: 3 * B].any()
    parity.check_mag(mag, o.compute_spectrogram(x, V), B, "zero frames")
    u8 = e.compute_spectrogram_u8(x, V, 20.0, 80.0)
    assert np.array_equal(u8[: 3 * B], o.compute_spectrogram_u8(x, V, 20.0, 80.0)[: 3 * B])
    assert np.array_equal(e.get_peaks(0.0), o.get_peaks(0.0))
    assert not e.get_peak_magnitudes(0.0)[:3].any()
    e.free()


@pytest.mark.parametrize("N,V", TILED)
def test_batch_streams_with_partial_tiles(N, V):
    # stream lengths that leave the last tile of every stream partly empty, whatever the frames-per-wave count
    import torch
    H, B, S = N - V, N // 2, 3
    F = 19                                                     # odd, and not a multiple of 4, 8 or 16
    per = N + (F - 1) * H + 1
    stride = per + 37
    buf = np.zeros(S * stride, np.float32)
    for s in range(S):
        buf[s * stride: s * stride + per] = signal(per, 300 + s)
    dx = torch.from_numpy(buf).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    o = so.OracleEngine(N, "hann")
    assert sa.frames_for(per, N, V) == F
    out = torch.full((S * F * B + 64,), 7, dtype=torch.uint8, device="cuda")      # 64 guard bytes behind the rows
    mag = torch.zeros(S * F * B, dtype=torch.float32, device="cuda")
    pb = torch.zeros(S * F, dtype=torch.int16, device="cuda")
    pm = torch.zeros(S * F, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e.stft_u8_dev(dx.data_ptr(), S, stride, per, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(),
                  gm.data_ptr(), 0)
    e.stft_mag_f32_dev(dx.data_ptr(), S, stride, per, V, mag.data_ptr(), 0)
    torch.cuda.synchronize()
    out, mag = out.cpu().numpy(), mag.cpu().numpy().reshape(S, -1)
    assert (out[S * F * B:] == 7).all()                        # nothing written past the last row
    out = out[: S * F * B].reshape(S, -1)
    pb = pb.cpu().numpy().view(np.uint16).reshape(S, F)
    for s in range(S):
        xs = buf[s * stride: s * stride + per]
        parity.check_mag(mag[s], o.compute_spectrogram(xs, V), B, f"stream {s}")
        nd, nt = parity.u8_diff(out[s], o.compute_spectrogram_u8(xs, V, 20.0, 80.0))
        assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
        assert np.array_equal(pb[s], o.get_peaks(0.0))
    e.free()


@pytest.mark.parametrize("N,V", [(1024, 768), (4096, 3891), (512, 0)])
def test_features_only_equals_the_u8_call(kernel_path, N, V):
    # sono_stft_features_dev = sono_stft_u8_dev minus the rows: same peaks, same global maximum, bit for bit
    import torch
    H, S = N - V, 3
    per = N + 41 * H + 5
    stride = per + 11
    buf = np.zeros(S * stride, np.float32)
    for s in range(S):
        buf[s * stride: s * stride + per] = signal(per, 700 + s)
    dx = torch.from_numpy(buf).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    F = sa.frames_for(per, N, V)
    out = torch.zeros(S * F * (N // 2), dtype=torch.uint8, device="cuda")
    pb = [torch.zeros(S * F, dtype=torch.int16, device="cuda") for _ in range(2)]
    pm = [torch.zeros(S * F, dtype=torch.float32, device="cuda") for _ in range(2)]
    gm = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    e.stft_u8_dev(dx.data_ptr(), S, stride, per, V, 20.0, 80.0, out.data_ptr(), pb[0].data_ptr(), pm[0].data_ptr(),
                  gm[0].data_ptr(), 0)
    e.stft_features_dev(dx.data_ptr(), S, stride, per, V, pb[1].data_ptr(), pm[1].data_ptr(), gm[1].data_ptr(), 0)
    e.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(pb[0], pb[1]) and torch.equal(pm[0], pm[1]) and torch.equal(gm[0], gm[1])
    assert pm[0].max().item() > 0
    with pytest.raises(sa.SonoError):
        e.stft_features_dev(dx.data_ptr(), S, stride, per, V, 0, 0, 0, 0)
    e.free()


@pytest.mark.parametrize("N,V", [(1024, 768), (512, 256), (4096, 3891), (256, 0)])
def test_ragged_batch_equals_one_call_per_stream(kernel_path, N, V):
    # streams of different lengths (one empty, one shorter than a frame, one of exactly one frame) in ONE launch
    import torch
    H, B = N - V, N // 2
    lengths = [N + 17 * H + 3, 0, N - 1, N, N + 40 * H, N + 5 * H + H - 1, N + 2 * H]
    gaps = [5, 0, 3, 1, 0, 8, 2]
    offsets, pos = [], 7
    for ln, g in zip(lengths, gaps):
        offsets.append(pos)
        pos += ln + g
    buf = np.zeros(pos + 16, np.float32)
    for i, (o, ln) in enumerate(zip(offsets, lengths)):
        buf[o:o + ln] = signal(ln, 900 + i) if ln else 0
    dx = torch.from_numpy(buf).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    row0 = sa.ragged_rows(lengths, N, V)
    assert row0.tolist() == np.concatenate([[0], np.cumsum([sa.frames_for(ln, N, V) for ln in lengths])]).tolist()
    T = int(row0[-1])
    out = torch.full((T * B + 64,), 7, dtype=torch.uint8, device="cuda")
    mag = torch.zeros(T * B, dtype=torch.float32, device="cuda")
    pb = torch.zeros(T, dtype=torch.int16, device="cuda")
    pm = torch.zeros(T, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    pb2, pm2, gm2 = torch.zeros_like(pb), torch.zeros_like(pm), torch.zeros_like(gm)
    torch.cuda.synchronize()
    e.stft_u8_ragged_dev(dx.data_ptr(), offsets, lengths, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(),
                         gm.data_ptr(), 0)
    e.stft_mag_f32_ragged_dev(dx.data_ptr(), offsets, lengths, V, mag.data_ptr(), 0)
    e.stft_u8_ragged_dev(dx.data_ptr(), offsets, lengths, V, 20.0, 80.0, 0, pb2.data_ptr(), pm2.data_ptr(), gm2.data_ptr(), 0)
    e.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(pb, pb2) and torch.equal(pm, pm2) and torch.equal(gm, gm2)           # features only = same features
    out_h, mag_h = out.cpu().numpy(), mag.cpu().numpy()
    assert (out_h[T * B:] == 7).all()
    pbh, pmh = pb.cpu().numpy().view(np.uint16), pm.cpu().numpy()
    best = 0.0
    for i, (o, ln) in enumerate(zip(offsets, lengths)):
        r0, r1 = int(row0[i]), int(row0[i + 1])
        xs = buf[o:o + ln]
        u8 = e.compute_spectrogram_u8(xs, V, 20.0, 80.0)                                     # same kernel: bit-exact
        assert np.array_equal(out_h[r0 * B:r1 * B], u8), i
        assert np.array_equal(mag_h[r0 * B:r1 * B], e.compute_spectrogram(xs, V)), i
        if r1 > r0:
            assert np.array_equal(pbh[r0:r1], e.get_peaks(0.0)) and np.array_equal(pmh[r0:r1], e.get_peak_magnitudes(0.0))
            best = max(best, e.get_global_max())
    assert gm.cpu().numpy().view(np.float32)[0] == np.float32(best)
    # an empty batch is not an error
    e.stft_u8_ragged_dev(dx.data_ptr(), [], [], V, 20.0, 80.0, out.data_ptr(), 0, 0, 0, 0)
    e.free()


def test_ragged_launches_back_to_back_keep_their_own_tables():
    # four ragged launches queued without a host synchronisation in between, each with another stream order
    import torch
    N, V = 1024, 768
    H, B = N - V, N // 2
    lengths = [N + (5 + 3 * i) * H for i in range(6)]
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    buf = np.concatenate([signal(ln, 950 + i) for i, ln in enumerate(lengths)])
    dx = torch.from_numpy(buf).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    single = [e.compute_spectrogram_u8(buf[int(o):int(o) + ln], V, 20.0, 80.0) for o, ln in zip(offsets, lengths)]
    orders = [[0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [2, 0, 4, 1, 5, 3], [3, 3, 0, 5, 1, 1]]
    outs = []
    st = torch.cuda.Stream()
    for order in orders:
        total = int(sa.ragged_rows([lengths[i] for i in order], N, V)[-1])
        outs.append(torch.zeros(total * B, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    for order, out in zip(orders, outs):
        e.stft_u8_ragged_dev(dx.data_ptr(), [offsets[i] for i in order], [lengths[i] for i in order], V, 20.0, 80.0,
                             out.data_ptr(), 0, 0, 0, st.cuda_stream)
    st.synchronize()
    for order, out in zip(orders, outs):
        assert np.array_equal(out.cpu().numpy(), np.concatenate([single[i] for i in order]))
    e.free()


@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096])
def test_shared_loads_change_nothing(monkeypatch, N):
    # hop = N/4: frame B takes the rows it shares with frame A from A's registers instead of loading them again;
    # the engine option "force_any_hop" selects the form that loads both frames whole -- identical bits
    V, H = N - N // 4, N // 4
    x = signal(N + 41 * H + 5, N + 1)
    x[: N + H] = 0.0
    x[N + 6 * H: N + 7 * H] *= 300.0                      # a burst that only some frames see
    res = []
    monkeypatch.delenv("SONO_FORCE_ANY_HOP", raising=False)
    for any_hop in (False, True):
        e = sa.SpectrogramEngine(N, "hann")
        assert not e.get_option("force_any_hop")
        e.set_option("force_any_hop", any_hop)
        u8 = e.compute_spectrogram_u8(x, V, 20.0, 80.0)
        res.append((e.compute_spectrogram(x, V), u8, e.get_peaks(0.0), e.get_peak_magnitudes(0.0), np.float32(e.get_global_max())))
        e.free()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert res[0][1].any()
