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
