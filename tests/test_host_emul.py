"""The generic STFT workgroup, emulated phase by phase on the CPU
(tests/host_emul), against the oracle.  This checks the kernel's index math and
butterflies in the no-GPU suite; the same code runs as the HIP kernel on the GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import sono_oracle as so
from tests import parity

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, "host_emul")


@pytest.fixture(scope="module")
def emul():
    if os.environ.get("SONO_EMUL_LIB"):                  # another build of the same sources (tools/asan_tests.sh)
        return C.CDLL(os.environ["SONO_EMUL_LIB"])
    lib = os.path.join(EMUL_DIR, "libemul.so")
    srcs = [os.path.join(EMUL_DIR, f) for f in os.listdir(EMUL_DIR) if f.endswith(".cpp")]
    deps = srcs + [os.path.join(HERE, "..", "spectrogram_amd", "csrc", h)
                   for h in ("sono_core.h", "sono_tables.h", "sono_tile_core.h", "sono_rowscan.h", "sono_measure_core.h")
                   if os.path.exists(os.path.join(HERE, "..", "spectrogram_amd", "csrc", h))]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-o", lib] + srcs)
    return C.CDLL(lib)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def run_emul(L, fn, mode, x, N, V, window, gain=20.0, rng=80.0, n_streams=1, stride=None):
    per = len(x) if stride is None else stride
    H, B = N - V, N // 2
    F = so.num_frames(per if n_streams > 1 else len(x), N, V)
    T = F * n_streams
    mag = np.zeros(T * B, np.float32)
    u8 = np.zeros(T * B, np.uint8)
    pb = np.zeros(T, np.uint16)
    pm = np.zeros(T, np.float32)
    gm = np.zeros(1, np.uint32)
    pw = np.zeros(T * (B + 1), np.float32)
    rc = getattr(L, fn)(C.c_int(mode), _p(x), C.c_ulonglong(n_streams), C.c_ulonglong(per), C.c_ulonglong(F),
                        C.c_uint(H), C.c_uint(N), _p(window), C.c_float(gain), C.c_float(rng),
                        _p(mag), _p(u8), _p(pb), _p(pm), _p(gm), _p(pw))
    assert rc == 0
    return dict(mag=mag, u8=u8, peak_bin=pb, peak_mag=pm, gmax=float(gm.view(np.float32)[0]), power=pw, F=F)


def signal(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    return (0.3 * np.sin(0.2 * t + 1e-5 * t * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)


@pytest.mark.parametrize("N,V", [(256, 128), (512, 256), (1024, 768), (2048, 1536), (4096, 3891), (1024, 0)])
def test_generic_block_matches_oracle(emul, N, V):
    x = signal(N + 9 * (N - V) + 5, N)
    e = so.OracleEngine(N, "hann")
    r = run_emul(emul, "emul_generic", 0, x, N, V, e.window)
    parity.check_mag(r["mag"], e.compute_spectrogram(x, V), N // 2, f"emul mag N={N}")
    r = run_emul(emul, "emul_generic", 1, x, N, V, e.window)
    ref = e.compute_spectrogram_u8(x, V, 20.0, 80.0)
    nd, nt = parity.u8_diff(r["u8"], ref)
    assert nd <= max(2, 1e-3 * nt)
    assert np.array_equal(r["peak_bin"], e.get_peaks(0.0))
    assert np.allclose(r["peak_mag"], e.get_peak_magnitudes(0.0), rtol=1e-6, atol=0)
    assert abs(r["gmax"] - e.get_global_max()) <= 1e-6 * e.get_global_max()


def test_generic_block_batch_layout(emul):
    # three streams with a stride larger than their length: rows land stream-major
    N, V, per, stride = 512, 384, 512 + 5 * 128, 2000
    e = so.OracleEngine(N, "hamming")
    buf = np.zeros(3 * stride, np.float32)
    refs = []
    for s in range(3):
        xs = signal(per, 100 + s)
        buf[s * stride: s * stride + per] = xs
        refs.append(e.compute_spectrogram(xs, V))
    H, B = N - V, N // 2
    F = so.num_frames(per, N, V)
    mag = np.zeros(3 * F * B, np.float32)
    rc = emul.emul_generic(C.c_int(0), _p(buf), C.c_ulonglong(3), C.c_ulonglong(stride), C.c_ulonglong(F),
                           C.c_uint(H), C.c_uint(N), _p(e.window), C.c_float(20), C.c_float(80),
                           _p(mag), None, None, None, None, None)
    assert rc == 0
    parity.check_mag(mag, np.concatenate(refs), B, "batch")


def test_generic_block_power_mode(emul):
    N, hop = 1024, 256
    x = signal(N + 6 * hop, 5) + np.float32(0.02)
    w = so.create_window("hann", N)
    r = run_emul(emul, "emul_generic", 2, x, N, N - hop, w)
    F, B1 = r["F"], N // 2 + 1
    p = r["power"].reshape(F, B1)
    s = np.zeros(B1, np.float32)
    for f in range(F):
        s += p[f]
    avg = s / np.float32(F)
    psd = (np.sqrt(avg) ** 2) / np.float32(N)
    db = 10 * np.log10(np.maximum(psd, 1e-16))
    parity.check_power(db.astype(np.float32), so.compute_power_spectrum(x, 384000, N, "hann", 75.0), "emul power")


@pytest.mark.parametrize("N,V", [(1024, 768), (1024, 512), (1024, 992), (1024, 0), (1024, 1022), (512, 256), (512, 384), (256, 128), (256, 192), (256, 64),
                                 (2048, 1536), (2048, 0), (2048, 2000), (4096, 3891), (4096, 2048), (4096, 0)])
def test_tile_wave_matches_oracle(emul, N, V):
    H = N - V
    x = signal(N + 10 * H + (H - 1 if H > 1 else 0), N + 1)
    e = so.OracleEngine(N, "hann")
    r = run_emul(emul, "emul_tile", 0, x, N, V, e.window)
    parity.check_mag(r["mag"], e.compute_spectrogram(x, V), N // 2, f"tile mag N={N}")
    r = run_emul(emul, "emul_tile", 1, x, N, V, e.window)
    ref = e.compute_spectrogram_u8(x, V, 20.0, 80.0)
    nd, nt = parity.u8_diff(r["u8"], ref)
    assert nd <= max(2, 1e-3 * nt)
    assert np.array_equal(r["peak_bin"], e.get_peaks(0.0))
    assert np.allclose(r["peak_mag"], e.get_peak_magnitudes(0.0), rtol=1e-6, atol=0)
    assert abs(r["gmax"] - e.get_global_max()) <= 1e-6 * e.get_global_max()


def test_tile_wave_batch_with_partial_tiles(emul):
    # 3 streams of 9 frames each: 9 is not a multiple of the 4 frames a wave holds at N=1024
    N, V, per, stride = 1024, 768, 1024 + 8 * 256, 4000
    e = so.OracleEngine(N, "hann")
    buf = np.zeros(3 * stride, np.float32)
    refs = []
    for s in range(3):
        xs = signal(per, 200 + s)
        buf[s * stride: s * stride + per] = xs
        refs.append(e.compute_spectrogram(xs, V))
    F = so.num_frames(per, N, V)
    mag = np.zeros(3 * F * 512, np.float32)
    rc = emul.emul_tile(C.c_int(0), _p(buf), C.c_ulonglong(3), C.c_ulonglong(stride), C.c_ulonglong(F),
                        C.c_uint(N - V), C.c_uint(N), _p(e.window), C.c_float(20), C.c_float(80),
                        _p(mag), None, None, None, None, None)
    assert rc == 0
    parity.check_mag(mag, np.concatenate(refs), 512, "tile batch")


@pytest.mark.parametrize("N,V", [(1024, 768), (512, 256), (256, 192), (2048, 1536), (4096, 3891)])
def test_tile_wave_quiet_frames_next_to_loud_ones(emul, N, V):
    # frames are transformed in pairs: a loud frame must not leak rounding noise into its quiet partner
    H = N - V
    rng = np.random.default_rng(3)
    n = N + 23 * H
    x = (1e-5 * rng.standard_normal(n)).astype(np.float32)
    t = np.arange(n)
    burst = (t > 9 * H) & (t < 9 * H + N // 2)
    x[burst] += (0.8 * np.sin(0.7 * t[burst])).astype(np.float32)
    e = so.OracleEngine(N, "hann")
    r = run_emul(emul, "emul_tile", 0, x, N, V, e.window)
    parity.check_mag(r["mag"], e.compute_spectrogram(x, V), N // 2, f"quiet/loud mag N={N}")
    r = run_emul(emul, "emul_tile", 1, x, N, V, e.window)
    nd, nt = parity.u8_diff(r["u8"], e.compute_spectrogram_u8(x, V, 20.0, 80.0))
    assert nd <= max(2, 1e-3 * nt)
    assert np.array_equal(r["peak_bin"], e.get_peaks(0.0))


@pytest.mark.parametrize("N,V", [(1024, 768), (2048, 1536), (4096, 3072)])
def test_tile_wave_all_zero_frames_stay_exactly_zero(emul, N, V):
    H = N - V
    x = np.zeros(N + 15 * H, np.float32)
    x[N + 2 * H:] = signal(len(x) - (N + 2 * H), 9)          # frames 0..2 see only zeros
    e = so.OracleEngine(N, "hann")
    r = run_emul(emul, "emul_tile", 0, x, N, V, e.window)
    ref = e.compute_spectrogram(x, V)
    B = N // 2
    assert not r["mag"][: 3 * B].any() and not ref[: 3 * B].any()
    parity.check_mag(r["mag"], ref, B, "zero frames")
    r = run_emul(emul, "emul_tile", 1, x, N, V, e.window)
    assert np.array_equal(r["u8"][: 3 * B], e.compute_spectrogram_u8(x, V, 20.0, 80.0)[: 3 * B])
    assert np.array_equal(r["peak_bin"], e.get_peaks(0.0))
    assert not r["peak_mag"][:3].any()


@pytest.mark.parametrize("N,V,log_r2", [(1024, 768, 6), (1024, 1000, 6), (512, 256, 6), (2048, 1536, 5), (2048, 1948, 5), (2048, 0, 5)])
def test_tile_wave_alternative_geometry(emul, N, V, log_r2):
    # N = 1024 / 512 through the 64-column geometry (R1 = 16 / 8), N = 2048 as 64 x 32 (one frame per lane in
    # pass 1, the pair's frames in the wave's two halves): same results within the tolerances
    H = N - V
    x = signal(N + 21 * H + 3, N + 2)
    x[: N + 2 * H] = 0.0
    e = so.OracleEngine(N, "hann")
    emul.emul_tile_set_log_r2(log_r2)
    try:
        r = run_emul(emul, "emul_tile", 0, x, N, V, e.window)
        parity.check_mag(r["mag"], e.compute_spectrogram(x, V), N // 2, f"wide tile mag N={N}")
        assert not r["mag"][: 3 * (N // 2)].any()
        r = run_emul(emul, "emul_tile", 1, x, N, V, e.window)
        nd, nt = parity.u8_diff(r["u8"], e.compute_spectrogram_u8(x, V, 20.0, 80.0))
        assert nd <= max(2, 1e-3 * nt)
        assert np.array_equal(r["peak_bin"], e.get_peaks(0.0))
    finally:
        emul.emul_tile_set_log_r2(0)


@pytest.mark.parametrize("N,V,lo,hi", [(1024, 512, 27, 341), (512, 256, 0, 255), (256, 128, 100, 100), (2048, 1024, 5, 1000),
                                       (4096, 2048, 300, 2047)])
def test_tile_wave_band_energy(emul, N, V, lo, hi):
    # the fused fast-scan mode: per-frame sum of (double)mag^2 over a band, against the oracle's magnitudes
    H = N - V
    x = signal(N + 12 * H + 3, N + 5)
    x[: N + H] = 0.0                                       # the first two frames see only zeros
    e = so.OracleEngine(N, "hann")
    F = so.num_frames(len(x), N, V)
    mags = e.compute_spectrogram(x, V).reshape(F, N // 2).astype(np.float64)
    want = (mags[:, lo:hi + 1] ** 2).sum(axis=1)
    target = float(np.median(want))
    active = np.full(F, 7, np.uint8)
    energy = np.zeros(F, np.float64)
    emul.emul_tile_set_band(C.c_uint(lo), C.c_uint(hi), C.c_double(target), _p(active), _p(energy))
    run_emul(emul, "emul_tile", 3, x, N, V, e.window)
    assert energy[0] == 0.0 and energy[1] == 0.0
    assert np.allclose(energy, want, rtol=3e-4, atol=0)    # f32 magnitudes agree to 1e-4
    assert np.array_equal(active, (energy > target).astype(np.uint8))


def test_tile_wave_survey_pass(emul):
    # features at hop N/4 and the band test on the even frames from one pass: peaks equal the row mode's, flags and sums
    # equal the band mode's at hop N/2 (the sums to the f32 magnitudes' last bits: another partner frame, another order)
    N, V, lo, hi = 1024, 768, 26, 341
    for n in (N + 13 * 256 + 7, N + 12 * 256):                 # 14 frames -> 7 scan frames; 13 -> 7
        x = signal(n, N + 9)
        x[: N + 2 * 256] = 0.0
        e = so.OracleEngine(N, "hann")
        F, FS = so.num_frames(n, N, V), so.num_frames(n, N, 512)
        assert FS == (F + 1) // 2
        want = run_emul(emul, "emul_tile", 1, x, N, V, e.window)
        ref_active, ref_energy = np.full(FS, 7, np.uint8), np.zeros(FS, np.float64)
        probe = np.zeros(FS, np.float64)
        emul.emul_tile_set_band(C.c_uint(lo), C.c_uint(hi), C.c_double(0.0), _p(ref_active), _p(probe))
        run_emul(emul, "emul_tile", 3, x, N, 512, e.window)
        srt = np.sort(probe)
        target = float(0.5 * (srt[FS // 2 - 1] + srt[FS // 2]))          # between two frames' sums, not on one
        emul.emul_tile_set_band(C.c_uint(lo), C.c_uint(hi), C.c_double(target), _p(ref_active), _p(ref_energy))
        run_emul(emul, "emul_tile", 3, x, N, 512, e.window)
        active, energy = np.full(FS, 9, np.uint8), np.zeros(FS, np.float64)
        emul.emul_tile_set_band(C.c_uint(lo), C.c_uint(hi), C.c_double(target), _p(active), _p(energy))
        got = run_emul(emul, "emul_tile", 4, x, N, V, e.window)
        assert np.array_equal(got["peak_bin"], want["peak_bin"]) and np.array_equal(got["peak_mag"], want["peak_mag"])
        assert got["gmax"] == want["gmax"] and not got["u8"].any()              # no rows in this mode
        assert np.array_equal(active, ref_active) and np.allclose(energy, ref_energy, rtol=1e-6, atol=0)
        assert energy[0] == 0.0 and 0 < active.sum() < FS


@pytest.mark.parametrize("V,nf,taps", [(768, 512, 2), (700, 64, 2), (0, 36, 3)])
def test_tile_wave_filter_bank_in_the_epilogue(emul, V, nf, taps):
    # apply_filter_bank (lib.rs:310-334) from the power image, against the oracle's rows
    N, B = 1024, 512
    rng = np.random.default_rng(nf + V)
    bank = np.zeros((nf, B + 1), np.float32)                   # rows of B + 1 weights, used with stride B (the reference's quirk)
    for i, p in enumerate(np.sort(rng.uniform(0, B - taps, nf))):
        if i % 5 == 4:
            continue                                           # filters without any tap
        o = int(p)
        w = rng.uniform(0.05, 1.0, taps)
        bank[i, o:o + taps] = (w / w.sum()).astype(np.float32)
    flat = np.ascontiguousarray(bank.ravel())
    H = N - V
    x = signal(N + 9 * H + 5, nf)
    x[: N + H] = 0.0
    o = so.OracleEngine(N, "hann")
    o.load_filter_bank(flat, nf)
    want = np.asarray(o.compute_spectrogram_u8(x, V, 20.0, 80.0))
    F = so.num_frames(len(x), N, V)
    emul.emul_tile_set_bank.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    emul.emul_tile_set_bank(_p(flat), flat.size, nf)
    got = run_emul(emul, "emul_tile", 5, x, N, V, o.window)
    rows = got["u8"][: F * nf]
    nd, nt = parity.u8_diff(rows, want, "bank in the epilogue")
    assert nt == F * nf and nd <= max(2, 1e-3 * nt)
    assert not rows[: 2 * nf].any() and rows.max() > 100
    assert np.array_equal(got["peak_bin"], np.asarray(o.get_peaks(0.0)))


@pytest.mark.parametrize("N,V", [(1024, 768), (256, 128), (4096, 3891)])
def test_tile_wave_ragged_tables_give_the_same_rows(emul, N, V):
    # the same batch walked through the ragged-batch tables (per-tile stream ids, per-stream offsets): identical output
    H, B = N - V, N // 2
    per, stride = N + 10 * H + 1, N + 12 * H
    buf = np.zeros(3 * stride, np.float32)
    for s in range(3):
        buf[s * stride: s * stride + per] = signal(per, 400 + s)
    e = so.OracleEngine(N, "hann")
    res = []
    for ragged in (0, 1):
        emul.emul_tile_set_ragged(ragged)
        try:
            res.append(_run_batch(emul, buf, N, V, e.window, per, stride))
        finally:
            emul.emul_tile_set_ragged(0)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][0].any()


def _run_batch(emul, buf, N, V, window, per, stride):
    F, B = so.num_frames(per, N, V), N // 2
    u8 = np.zeros(3 * F * B, np.uint8)
    pb = np.zeros(3 * F, np.uint16)
    rc = emul.emul_tile(C.c_int(1), _p(buf), C.c_ulonglong(3), C.c_ulonglong(stride), C.c_ulonglong(F), C.c_uint(N - V),
                        C.c_uint(N), _p(window), C.c_float(20), C.c_float(80), None, _p(u8), _p(pb), None, None, None)
    assert rc == 0
    return u8, pb


@pytest.mark.parametrize("N,V,mode", [(1024, 768, 1), (1024, 768, 0), (512, 384, 1), (256, 192, 1), (2048, 1536, 1), (4096, 3072, 0),
                                      (1024, 512, 3), (2048, 1024, 3)])
def test_tile_wave_shared_loads_change_nothing(emul, N, V, mode):
    # hop = N/4 (N/2 in the band mode): frame B takes the rows it shares with frame A from A's registers
    # instead of loading them again -- every output bit must equal the form that loads both frames whole
    H = N - V
    x = signal(N + 13 * H + 5, N + 9)
    x[: N + H] = 0.0
    x[N + 6 * H: N + 7 * H] *= 300.0                      # a burst that only some frames see (frame B's scaling)
    e = so.OracleEngine(N, "hann")
    res = []
    for any_hop in (0, 1):
        emul.emul_tile_set_any_hop(any_hop)
        try:
            if mode == 3:
                F = so.num_frames(len(x), N, V)
                active = np.full(F, 7, np.uint8)
                energy = np.zeros(F, np.float64)
                emul.emul_tile_set_band(C.c_uint(3), C.c_uint(N // 2 - 2), C.c_double(1e-3), _p(active), _p(energy))
                run_emul(emul, "emul_tile", 3, x, N, V, e.window)
                res.append((active, energy))
            else:
                r = run_emul(emul, "emul_tile", mode, x, N, V, e.window)
                res.append((r["mag"], r["u8"], r["peak_bin"], r["peak_mag"], np.float32(r["gmax"])))
        finally:
            emul.emul_tile_set_any_hop(0)
    for u, v in zip(res[0], res[1]):
        assert np.array_equal(u, v)
    assert any(np.any(u) for u in res[0])


def test_tile_peak_rule_on_near_ties(emul):
    """The kernels take the per-frame peak on |2X|^2: the largest power wins, the lowest bin among EQUAL powers.  The
    reference takes it on the rounded f32 magnitude, first maximum wins (lib.rs:370-378).  The two rules differ only when
    two powers an ulp or two apart round to the same magnitude and the earlier bin holds the smaller power -- the case
    built here; the magnitude reported is the same either way."""
    import ctypes as C
    emul.emul_peak_of_powers.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint), C.POINTER(C.c_float)]
    emul.emul_peak_of_powers.restype = C.c_int

    def peak(p):
        p = np.ascontiguousarray(p, np.float32)
        b, m = C.c_uint(0), C.c_float(0)
        assert emul.emul_peak_of_powers(p.ctypes.data_as(C.POINTER(C.c_float)), C.byref(b), C.byref(m)) == 0
        return b.value, np.float32(m.value)

    def reference_rule(p):                    # lib.rs:370-378 on mag = sqrt(p) / N in f32
        mag = np.sqrt(p.astype(np.float32)) * np.float32(1.0 / 1024.0)
        best, at = np.float32(0.0), 0
        for k, v in enumerate(mag):
            if v > best:
                best, at = v, k
        return at, best

    rng = np.random.default_rng(3)
    base = (rng.random(512) * 0.5 + 0.1).astype(np.float32)
    # exact ties: the lowest bin wins under both rules, wherever the tied bins sit (same lane, other lanes, other groups)
    for bins in ([7, 300], [300, 301], [4, 5, 6, 7], [0, 511], [129, 130, 400]):
        p = base.copy()
        p[bins] = np.float32(2.5)
        assert peak(p)[0] == reference_rule(p)[0] == min(bins)
    # a near-tie: two powers one ulp apart whose square roots round to the same f32
    lo = np.float32(2.5)
    hi = np.nextafter(lo, np.float32(4.0))
    assert hi > lo and np.sqrt(lo) == np.sqrt(hi)
    p = base.copy()
    p[10], p[200] = lo, hi
    kb, km = peak(p)
    rb, rm = reference_rule(p)
    assert (kb, rb) == (200, 10)              # the larger power / the first of the equal magnitudes
    assert km == rm                           # and the same magnitude
    p[10], p[200] = hi, lo                    # the earlier bin larger: both rules agree
    assert peak(p)[0] == reference_rule(p)[0] == 10
    assert peak(np.zeros(512, np.float32)) == (0, np.float32(0.0))


def test_quantiser_rounding_rule_is_pinned(emul):
    """The reference truncates v = (dB - floor) * 255 / range to a code (lib.rs:277-293: `as u8`).  The kernels convert
    v - 1/2 with round-to-nearest-even (one v_cvt_pk_u8_f32), which equals trunc(v) for every v >= 0 that is not an
    integer, and differs in exactly one place: an exactly integral ODD v comes out one code lower (v - 1/2 is a tie and
    goes to the even neighbour below).  That deviation is deliberate and is what the 1-LSB tolerance of the u8 rows
    covers; this test pins it so that a later change cannot widen it silently."""
    emul.emul_code_of.restype = C.c_uint
    emul.emul_code_of.argtypes = [C.c_float]
    code = lambda v: emul.emul_code_of(C.c_float(v))
    for k in range(0, 256):
        assert code(float(k)) == (k if k % 2 == 0 else k - 1), k                   # exact integers: even stays, odd drops one
        if k < 255:
            for frac in (2.0 ** -16, 0.25, 0.5, 0.75, 1.0 - 2.0 ** -16):           # everything in between truncates
                assert code(k + frac) == k, (k, frac)
    assert code(-0.0) == 0 and code(-1e-6) == 0 and code(-1e30) == 0 and code(float("-inf")) == 0      # saturation below
    assert code(255.4) == 255 and code(256.0) == 255 and code(1e30) == 255 and code(float("inf")) == 255      # and above
    assert code(float("nan")) == 0
    emul.emul_quantise.restype = C.c_uint
    emul.emul_quantise.argtypes = [C.c_float, C.c_uint, C.c_float, C.c_float]
    q = lambda p, N=1024, gain=20.0, rng=80.0: emul.emul_quantise(C.c_float(p), N, C.c_float(gain), C.c_float(rng))
    assert q(0.0) == 0 and q(float("nan")) == 0 and q(float("inf")) == 255 and q(1e-30) == 0
    assert q(0.0, 1024, 20.0, 200.0) == q(1e-30, 1024, 20.0, 200.0)                 # the 1e-10 floor, where the range can see it
    # full scale: |X| * 2/N = 1 -> 0 dB = gain above the top of the range -> 255; the top code's lower edge: -gain dB
    assert q((1.0 * 1024) ** 2) == 255
    top = (10.0 ** (-20.0 / 20.0) * 1024) ** 2
    assert q(top * 1.001) == 255 and q(top * 0.98) == 254


def test_row_scans_answer_what_the_scalar_loops_answer(emul):
    """sono_rowscan.h (four entries per comparison) against the loops of the reference it stands for
    (batCallDetector.js:2509-2522, 2002-2038, 3004-3020): rows with NaNs, infinities, signed zeros, ties, every length
    from 0 to 40 and every sub-range, thresholds given as doubles that are not floats."""
    L = emul
    L.rowscan_float_below.restype = C.c_float
    L.rowscan_float_below.argtypes = [C.c_double]
    L.rowscan_row_max.restype = C.c_float
    L.rowscan_row_max.argtypes = [C.c_void_p, C.c_ulonglong]
    for f in (L.rowscan_first_gt, L.rowscan_last_gt):
        f.restype = C.c_longlong
        f.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_float]
    L.rowscan_first_max.restype = C.c_longlong
    L.rowscan_first_max.argtypes = [C.c_void_p, C.c_ulonglong, C.c_void_p]
    rng = np.random.default_rng(5)
    # the float handed to the scans decides exactly as the double threshold would
    for thr in list(rng.normal(-60, 30, 200)) + [-57.25, 0.0, -0.0, 1e300, -1e300, float("inf"), float("-inf"), 3.0000001, -3.0000001]:
        tf = L.rowscan_float_below(thr)
        with np.errstate(over="ignore"):
            probe = np.float32(thr)
        for x in (probe, np.nextafter(probe, np.float32(np.inf)), np.nextafter(probe, np.float32(-np.inf))):
            assert (float(x) > thr) == (float(x) > tf), (thr, x, tf)
    assert np.isnan(L.rowscan_float_below(float("nan")))
    specials = np.asarray([np.nan, np.inf, -np.inf, 0.0, -0.0, -40.0, -40.0, -39.999996], np.float32)
    checked = 0
    for trial in range(400):
        n = int(rng.integers(0, 41))
        row = np.round(rng.normal(-50, 8, n)).astype(np.float32)                 # rounded: plenty of ties
        for k in range(int(rng.integers(0, 4))):
            if n:
                row[rng.integers(0, n)] = specials[rng.integers(0, len(specials))]
        if trial % 7 == 0:
            row[:] = np.nan
        if trial % 11 == 0:
            row[:] = -np.inf
        row = np.ascontiguousarray(row)
        # first maximum, as "if (v > best) {best = v; at = b}" from best = -inf leaves it
        best, at = -np.inf, -1
        for b, v in enumerate(row):
            if v > best:
                best, at = v, b
        val = np.zeros(1, np.float32)
        got = L.rowscan_first_max(_p(row), n, _p(val))
        assert got == at and (val[0] == np.float32(best)), (row, got, at)
        if at >= 0:
            assert np.signbit(val[0]) == np.signbit(row[at])
        m = L.rowscan_row_max(_p(row), n)
        assert m == np.float32(best)
        for t in (-50.0, -45.5, float(row[n // 2]) if n else 0.0, -np.inf, np.inf, np.nan):
            t32 = np.float32(t)
            for lo, hi in [(0, n - 1), (n // 3, n - 1), (0, n // 2), (n // 4, (3 * n) // 4), (2, 1), (n - 1, n - 1)]:
                if lo < 0 or hi >= n:
                    continue
                above = [b for b in range(lo, hi + 1) if row[b] > t32]
                assert L.rowscan_first_gt(_p(row), lo, hi, t32) == (above[0] if above else -1)
                assert L.rowscan_last_gt(_p(row), lo, hi, t32) == (above[-1] if above else -1)
                checked += 1
    assert checked > 5000


def test_pow15_is_correctly_rounded(emul):
    """x^1.5 of the contour's curvature (Math.pow(1 + d1*d1, 1.5), batCallDetector.js:3676) is worked out as x * sqrt(x)
    in two doubles each and rounded once, the same on the host and on the GPU.  Against 60-digit decimal arithmetic it
    is the correctly rounded double for every argument tried; the C library's pow is not (a unit in the last place off
    on a few arguments in ten thousand), which is why it is not what both executors call."""
    import math
    from decimal import Decimal, getcontext
    from fractions import Fraction
    getcontext().prec = 60
    L = emul
    L.measure_pow15.restype = C.c_double
    L.measure_pow15.argtypes = [C.c_double]
    rng = np.random.default_rng(17)
    xs = [1.0, 2.0, 4.0, 1.0 + 2.0 ** -52, 1e-3, 123456.789]
    xs += list(1.0 + np.exp(rng.uniform(-20, 10, 4000)) ** 2)                  # 1 + d1^2 over the slopes a contour can have
    xs += list(1.0 + rng.uniform(0, 1e-3, 2000)) + list(np.exp(rng.uniform(0, 400, 1000)))
    libm_off = 0
    for x in xs:
        x = float(x)
        exact = Decimal(x) * Decimal(x).sqrt()
        got = L.measure_pow15(x)
        # correctly rounded: no double lies strictly between got and the exact value's other side, i.e. |got - exact| <= ulp/2
        ulp = Decimal(math.ulp(got))
        assert abs(Decimal(got) - exact) <= ulp / 2, (x, got)
        libm_off += abs(Decimal(math.pow(x, 1.5)) - exact) > ulp / 2
    assert math.isinf(L.measure_pow15(float("inf"))) and math.isnan(L.measure_pow15(float("nan")))
    assert L.measure_pow15(1e250) == float("inf") or L.measure_pow15(1e250) > 1e300
    print("libm pow off by a unit on", libm_off, "of", len(xs))
