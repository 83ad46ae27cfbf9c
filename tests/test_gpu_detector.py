"""Detector front end on the GPU (spectrogram_amd.detector -> libsonostft)
against the golden vectors of the reference's JavaScript and the numpy oracle."""
import numpy as np
import pytest

import spectrogram_amd as sa
from oracle import detector_oracle as do
from oracle import sono_oracle as so
from spectrogram_amd import detector as det
from tests.test_detector_oracle import G, META, pairs

pytestmark = pytest.mark.gpu


def seg_pairs(segs, a, b):
    return [(s[a], s[b]) for s in segs]


@pytest.mark.parametrize("c", META, ids=[c["name"] for c in META])
def test_detector_matches_reference(c):
    n = c["name"]
    x = G[n + "/x"]
    e = sa.SpectrogramEngine(c["fft"], c["window"])
    segs = det.fast_scan_segments(e, x, c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"])
    assert seg_pairs(segs, "start", "end") == pairs(G[n + "/scan"])                     # call boundaries: exact
    merged = det.merge_and_pad_segments(segs, len(x), c["sr"], c["padding_ms"])
    assert seg_pairs(merged, "startSample", "endSample") == pairs(G[n + "/merged"])
    sp = det.generate_spectrogram(e, x, c["sr"], c["flow_khz"], c["fhigh_khz"], c["hop_percent"])
    F, nb = (int(v) for v in G[n + "/shape"])
    if F == 0:
        assert sp is None
        e.free()
        return
    assert sp["powerMatrix"].shape == (F, nb) and sp["powerMatrix"].dtype == np.float32
    ref = G[n + "/db"].reshape(F, nb)
    live = ref > ref.max(axis=1, keepdims=True) - 70.0
    assert np.abs(sp["powerMatrix"] - ref)[live].max() <= 0.01                          # dB
    assert np.array_equal(sp["timeFrames"], G[n + "/time"]) and np.array_equal(sp["freqBins"], G[n + "/freq"])
    assert sp["freqResolution"] == G[n + "/freqres"][0]
    calls = det.detect_call_segments(e, sp["powerMatrix"], c["call_threshold_db"])
    assert seg_pairs(calls, "startFrame", "endFrame") == pairs(G[n + "/calls"])
    calls = det.detect_call_segments(e, ref, c["call_threshold_db"])
    assert seg_pairs(calls, "startFrame", "endFrame") == pairs(G[n + "/calls"])
    e.free()


def test_frame_features_match_oracle():
    c = next(m for m in META if m["name"] == "pulses384k")
    F, nb = (int(v) for v in G["pulses384k/shape"])
    db = G["pulses384k/db"].reshape(F, nb)
    e = sa.SpectrogramEngine(1024, "hann")
    peak = float(db.max())
    for (b0, b1, lmax, lthr, hmin, hthr) in [(0, nb - 1, nb - 1, peak - 27.0, 0, peak - 24.0),
                                            (10, 150, 120, peak - 40.0, 30, peak - 10.0),
                                            (50, 40, 0, -1000.0, nb - 1, 1000.0)]:
        got = det.frame_features(e, db, b0, b1, lmax, lthr, hmin, hthr)
        ref = do.frame_features(db, b0, b1, lmax, lthr, hmin, hthr)
        for a, b in zip(got, ref):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    e.free()


def test_runs_edge_cases():
    e = sa.SpectrogramEngine(256, "hann")
    rng = np.random.default_rng(7)
    for F in (1, 2, 63, 64, 65, 1000, 4097):
        for kind in ("none", "all", "alternate", "random"):
            active = {"none": np.zeros(F, bool), "all": np.ones(F, bool), "alternate": np.arange(F) % 2 == 0,
                      "random": rng.random(F) < 0.3}[kind]
            # a dB matrix whose row maximum is 0 where active and -50 elsewhere; one loud row pins the global max
            db = np.full((F + 1, 5), -80.0, np.float32)
            db[:F, 2] = np.where(active, 0.0, -50.0)
            db[F, 1] = 10.0
            got = det.detect_call_segments(e, db, -24.0)
            want = do.runs(list(active) + [True])
            assert seg_pairs(got, "startFrame", "endFrame") == want, (F, kind)
    assert det.detect_call_segments(e, np.zeros((0, 4), np.float32)) == []
    e.free()


def test_merge_and_pad_is_host_only_and_sorts():
    segs = [{"start": 5000, "end": 6000}, {"start": 100, "end": 300}, {"start": 6100, "end": 6500}]
    got = det.merge_and_pad_segments(segs, 7000, 1000.0, 50.0)
    assert seg_pairs(got, "startSample", "endSample") == do.merge_and_pad_segments(
        [(s["start"], s["end"]) for s in segs], 7000, 1000.0, 50.0)
    assert det.merge_and_pad_segments([], 10, 1.0, 1.0) == []


def test_front_end_pipeline_matches_oracle_composition():
    # fast scan -> merge & pad -> per ROI spectrogram -> call segments -> segment peaks, GPU vs the pinned oracle stages
    from oracle.make_golden import chirp
    sr, flow, fhigh = 384000, 10.0, 128.0
    x = chirp(120000, sr, 91, period=0.06, t0=0.01)
    e = sa.SpectrogramEngine(1024, "hann")
    o = so.OracleEngine(1024, "hann")
    got = det.process_full_file_front_end(e, x, sr, flow, fhigh, threshold_db=-60, padding_ms=10)
    want = []
    merged = do.merge_and_pad_segments(do.fast_scan_segments(o, x, sr, flow, fhigh, -60), len(x), sr, 10)
    for roi, (s0, s1) in enumerate(merged):
        sp = do.generate_spectrogram(o, x[s0:s1], sr, flow, fhigh, 3.125)
        for f0, f1 in do.detect_call_segments(sp["powerMatrix"], -24.0):
            seg = sp["powerMatrix"][f0:f1 + 1]
            pf, pb = np.unravel_index(int(np.argmax(seg)), seg.shape)
            want.append((roi, s0, f0, f1, f0 + int(pf), float(sp["freqBins"][pb]) / 1000))
    assert len(got) == len(want) >= 5
    for g, w in zip(got, want):
        assert (g["roi"], g["startSample"], g["startFrame"], g["endFrame"], g["peakFrame"]) == w[:5]
        assert g["peakFreq_kHz"] == w[5]
        assert abs(g["duration_ms"] - (g["endFrame"] - g["startFrame"]) * 32 / sr * 1000) < 1e-9
    e.free()


@pytest.mark.parametrize("N,lo,hi", [(1024, 27, 341), (512, 0, 255), (2048, 100, 900), (4096, 10, 2047), (256, 3, 3)])
def test_fused_band_energy_equals_the_two_pass_form(N, lo, hi):
    # sono_stft_band_active_dev (no magnitude matrix) against sono_stft_mag_f32_dev + sono_detect_band_active_dev
    import ctypes as C
    import torch
    V, S = N // 2, 3
    per = N + 57 * (N - V) + 9
    stride = per + 5
    rng = np.random.default_rng(N)
    buf = np.zeros(S * stride, np.float32)
    for s in range(S):
        t = np.arange(per)
        buf[s * stride: s * stride + per] = (0.2 * np.sin(0.3 * t + 2e-5 * t * t) * (rng.random(per) > 0.3)
                                             + 0.01 * rng.standard_normal(per)).astype(np.float32)
    buf[: N + 3 * (N - V)] = 0.0
    dx = torch.from_numpy(buf).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    F = sa.frames_for(per, N, V)
    T, B = S * F, N // 2
    mags = torch.zeros(T * B, dtype=torch.float32, device="cuda")
    act = [torch.full((T,), 9, dtype=torch.uint8, device="cuda") for _ in range(2)]
    en = [torch.zeros(T, dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    e.stft_mag_f32_dev(dx.data_ptr(), S, stride, per, V, mags.data_ptr(), 0)
    e.synchronize()
    m = mags.cpu().numpy().reshape(T, B).astype(np.float64)
    want = (m[:, lo:hi + 1] ** 2).sum(axis=1)
    target = float(np.median(want))
    sa._check(sa.lib().sono_detect_band_active_dev(e._handle(), mags.data_ptr(), T, B, lo, hi, target, act[0].data_ptr(),
                                                   en[0].data_ptr(), None))
    e.stft_band_active_dev(dx.data_ptr(), S, stride, per, V, lo, hi, target, act[1].data_ptr(), en[1].data_ptr(), 0)
    e.synchronize()
    torch.cuda.synchronize()
    e0, e1 = en[0].cpu().numpy(), en[1].cpu().numpy()
    assert np.allclose(e1, want, rtol=1e-12, atol=0) and np.allclose(e0, e1, rtol=1e-12, atol=0)    # same terms, other order
    assert e1[0] == 0.0 and e1[2] == 0.0
    assert torch.equal(act[0], act[1]) and 0 < int(act[1].sum()) < T
    with pytest.raises(sa.SonoError):
        e.stft_band_active_dev(dx.data_ptr(), S, stride, per, V, lo, B, target, act[1].data_ptr(), 0, 0)
    e.free()


def test_db_matrix_is_the_library_logarithm_rounded_to_f32():
    """dB = f32(10 * log10(max(mag^2 / N, 1e-16))) with the logarithm in double (batCallDetector.js:1353-1360).  The
    kernel takes a short way to the logarithm and falls back to the long one wherever the f32 could differ: against
    numpy's double logarithm the f32 is the same everywhere, except where that double lies within 1e-12 dB of a rounding
    boundary of f32 (then either neighbour is what a double logarithm, a unit in the last place off, may give)."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(21)
    F, bins, min_bin, nb = 4096, 512, 27, 316
    mags = np.exp(rng.uniform(-45, 12, (F, bins))).astype(np.float32)
    mags[0, :] = 0.0
    mags[1, :] = np.float32(1e-42)                                           # denormal
    mags[2, 30:40] = np.nan
    mags[3, 30:40] = np.inf
    mags[4, :] = np.float32(32.0) * (1 + rng.uniform(-3e-6, 3e-6, bins)).astype(np.float32)        # psd next to 1: 0 dB
    mags[5, :] = np.float32(3.2e-7) * (1 + rng.uniform(-1e-3, 1e-3, bins)).astype(np.float32)      # psd next to the 1e-16 floor
    mags[6, :] = np.ldexp(np.float32(1.0), rng.integers(-60, 20, bins)).astype(np.float32)         # exact powers of two
    e = sa.SpectrogramEngine(1024, "hann")
    d_m = torch.from_numpy(mags).cuda()
    d_db = torch.empty((F, nb), dtype=torch.float32, device="cuda")
    rc = sa.lib().sono_detect_power_db_dev(e._handle(), C.c_void_p(d_m.data_ptr()), F, bins, min_bin, nb, C.c_void_p(d_db.data_ptr()), None)
    assert rc == 0
    e.synchronize()
    got = d_db.cpu().numpy()
    m = mags[:, min_bin:min_bin + nb].astype(np.float64)
    psd = (m * m) / 1024.0
    psd = np.where(psd > 1e-16, psd, 1e-16)
    with np.errstate(divide="ignore"):
        y = 10.0 * np.log10(psd)
    want = y.astype(np.float32)
    differ = got.view(np.uint32) != want.view(np.uint32)
    ambiguous = (y - 1e-12).astype(np.float32) != (y + 1e-12).astype(np.float32)
    assert not (differ & ~ambiguous).any(), np.argwhere(differ & ~ambiguous)[:5]
    assert differ.sum() <= 4 and (got[2, 3:13] == -160.0).all() and np.isinf(got[3, 3:13]).all() and (got[0] == -160.0).all()
    e.free()


@pytest.mark.parametrize("min_bin,nb", [(26, 316), (0, 512), (3, 64), (100, 65), (5, 1)])
def test_frame_peak_records_equal_in_both_forms(min_bin, nb):
    """sono_detect_frame_peaks_dev: the dB matrix + row pass (form 0) and the one-pass form the region analysis uses (1)
    leave the same records, bit for bit, on rows built around the places where the two could part: exact ties and
    near-ties of the largest magnitude (neighbouring floats, which round to one dB value: the first bin wins),
    magnitudes on both sides of 10^(-12/20) of the largest one float apart, peaks at the ends of a chunk of 64 bins and
    of the band, quiet rows down to the floor of the dB law, all-zero rows, and rows that hold a NaN, an infinity or a
    negative number."""
    import torch
    bins, N = 512, 1024
    rng = np.random.default_rng(nb * 1000 + min_bin)
    rows = []

    def noise(scale=1e-3):
        return (rng.rayleigh(scale, bins)).astype(np.float32)

    spots = sorted({0, 1, nb // 2, nb - 1, min(63, nb - 1), min(64, nb - 1), min(65, nb - 1), min(127, nb - 1), min(128, nb - 1)})
    for b in spots:                                             # a clear peak at the ends of chunks and of the band
        r = noise(); r[min_bin + b] = 0.3; rows.append(r)
    for b in spots:                                             # exact ties with a later bin, near-ties one to 40 floats below
        for ulps in (0, 1, 2, 3, 5, 8, 13, 40):
            r = noise(); r[min_bin + b] = 0.25
            other = min_bin + (b + 7) % nb
            r[other] = np.float32(0.25) if ulps == 0 else np.nextafter(np.float32(0.25), np.float32(0), dtype=np.float32) - np.float32((ulps - 1) * 2.0 ** -26)
            rows.append(r)
    ratio = np.float32(10.0 ** (-12.0 / 20.0))
    for top in (0.5, 0.123456, 3.3e-3):                         # magnitudes around peak - 12 dB, float by float
        edge = np.float32(np.float32(top) * ratio)
        for k in range(-6, 7):
            r = noise(1e-6); r[min_bin + nb // 3] = np.float32(top)
            v = edge
            for _ in range(abs(k)):
                v = np.nextafter(v, np.float32(1.0 if k > 0 else 0.0), dtype=np.float32)
            r[min_bin + (nb // 3 + 70) % nb] = v
            r[min_bin + (nb // 3 + 5) % nb] = v if nb > 8 else r[min_bin + (nb // 3 + 5) % nb]
            rows.append(r)
    for scale in (1e-5, 1e-6, 3e-7, 1e-7, 3e-8, 1e-8, 1e-9, 1e-12):   # quiet rows: the floor of the law comes within 12 dB of the peak
        rows.append(noise(scale)); r = noise(scale); r[min_bin + nb // 2] = np.float32(scale * 8); rows.append(r)
    rows.append(np.zeros(bins, np.float32))
    for bad in (np.nan, np.inf, -1.0):
        r = noise(); r[min_bin + nb // 2] = 0.2; r[min_bin + (nb // 2 + 3) % nb] = np.float32(bad); rows.append(r)
        r = noise(); r[min_bin + nb // 2] = np.float32(bad); rows.append(r)
    for _ in range(4000):                                       # and plain rows, with and without a tone
        r = noise(10.0 ** rng.uniform(-5, -1))
        if rng.random() < 0.7:
            r[min_bin + rng.integers(0, nb)] = np.float32(10.0 ** rng.uniform(-4, 0))
        rows.append(r)
    mags = torch.from_numpy(np.stack(rows)).cuda()
    F = mags.shape[0]
    e = sa.SpectrogramEngine(N, "hann")
    rec = np.dtype([("bin", "<i4"), ("db", "<f4"), ("left", "<f4"), ("right", "<f4"), ("lo", "<i4"), ("hi", "<i4")])
    got = []
    for form in (0, 1):
        out = torch.zeros(F * 6, dtype=torch.int32, device="cuda")
        sa._check(sa.lib().sono_detect_frame_peaks_dev(e._handle(), mags.data_ptr(), F, bins, min_bin, nb, out.data_ptr(), form, None))
        torch.cuda.synchronize()
        got.append(out.cpu().numpy().view(rec))
    for form in (1,):
        same = got[form].view(np.int32).reshape(F, 6) == got[0].view(np.int32).reshape(F, 6)
        bad_rows = np.flatnonzero(~same.all(axis=1))
        assert bad_rows.size == 0, (form, bad_rows[:5], got[form][bad_rows[:3]], got[0][bad_rows[:3]])
    # the records mean what they say: the peak is the first largest dB value of the row
    db = 10 * np.log10(np.maximum(mags.cpu().numpy()[:, min_bin:min_bin + nb].astype(np.float64) ** 2 / N, 1e-16))
    clean = np.isfinite(db).all(axis=1)
    assert np.abs(got[1]["db"][clean] - np.nanmax(db[clean], axis=1)).max() < 1e-4
    e.free()
