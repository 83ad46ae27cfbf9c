"""The survey-scale pass (a night-long channel per GPU: per-frame features, the fused fast scan and run
extraction, all on 16-bit PCM as it lies in a WAV data chunk) on ONE HOUR of a 384 kHz int16 channel:
the PCM forms against the float forms over every frame, the fused fast scan against the two-pass form
(magnitudes, then sono_detect_band_active_dev) on spot ranges, and the run lists against runs taken
from the flags with torch."""
import numpy as np
import pytest

import spectrogram_amd as sa
from spectrogram_amd import synth

pytestmark = pytest.mark.gpu

N, SR = 1024, 384000
SAMPLES = 3600 * SR
LO, HI = 26, 341                         # bins of 10 .. 128 kHz at 375 Hz per bin (batCallDetector.js:884-885)


@pytest.fixture(scope="module")
def night():
    import torch
    x = synth.chirp_torch(0, SAMPLES, SR, device="cuda")
    pcm = torch.clamp(torch.round(x * 32767.0), -32768, 32767).to(torch.int16)
    del x
    xf = pcm.to(torch.float32) / 32768.0                  # what the browser's decoder makes of the integers
    e = sa.SpectrogramEngine(N, "hann")
    torch.cuda.synchronize()
    yield dict(torch=torch, pcm=pcm, xf=xf, e=e)
    e.free()


def test_features_from_pcm16_equal_features_from_floats(night):
    torch, e = night["torch"], night["e"]
    V = 768
    F = sa.frames_for(SAMPLES, N, V)
    got, ref = [], []
    for fn, src in ((lambda *a: e.stft_u8_pcm16_dev(night["pcm"].data_ptr(), 1, SAMPLES, SAMPLES, 1, 0, V, 20.0, 80.0, 0, *a), got),
                    (lambda *a: e.stft_features_dev(night["xf"].data_ptr(), 1, SAMPLES, SAMPLES, V, *a), ref)):
        pb = torch.empty(F, dtype=torch.int16, device="cuda")
        pm = torch.empty(F, dtype=torch.float32, device="cuda")
        gm = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        fn(pb.data_ptr(), pm.data_ptr(), gm.data_ptr())
        e.synchronize()
        src.extend([pb, pm, gm])
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
    assert got[2].view(torch.float32).item() == got[1].max().item() > 0
    # a frame that starts 2 hops into a pulse (one per 100 ms = 150 frames) peaks on the 80 -> 30 kHz sweep
    on_sweep = got[0][2::150].to(torch.int32) & 0xFFFF
    assert on_sweep.min().item() >= 75 and on_sweep.max().item() <= 220


def test_fused_fast_scan_on_pcm16_and_the_run_lists(night):
    torch, e = night["torch"], night["e"]
    V = N // 2                                                            # 50 % overlap, js:880
    hop = N - V
    F = sa.frames_for(SAMPLES, N, V)
    energy = torch.empty(F, dtype=torch.float64, device="cuda")
    flags = torch.empty(F, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    e.stft_band_active_dev(night["xf"].data_ptr(), 1, SAMPLES, SAMPLES, V, LO, HI, 0.0, flags.data_ptr(), energy.data_ptr())
    e.synchronize()
    target = float(torch.quantile(energy[::97].to(torch.float32), 0.9).item())       # about one frame in ten is active
    ref_flags = torch.empty_like(flags)
    ref_energy = torch.empty_like(energy)
    e.stft_band_active_dev(night["xf"].data_ptr(), 1, SAMPLES, SAMPLES, V, LO, HI, target, ref_flags.data_ptr(), ref_energy.data_ptr())
    e.stft_band_active_pcm16_dev(night["pcm"].data_ptr(), 1, SAMPLES, SAMPLES, 1, 0, V, LO, HI, target, flags.data_ptr(),
                                 energy.data_ptr())
    e.synchronize()
    assert torch.equal(flags, ref_flags) and torch.equal(energy, ref_energy)
    active = int(flags.sum().item())
    assert 0.02 * F < active < 0.3 * F, active / F
    # the two-pass form on spot ranges that start on a tile boundary (so every frame meets the same partner)
    cnt = 40000
    mags = torch.empty(cnt * (N // 2), dtype=torch.float32, device="cuda")
    two = torch.empty(cnt, dtype=torch.uint8, device="cuda")
    two_energy = torch.empty(cnt, dtype=torch.float64, device="cuda")
    for f0 in (0, 16 * 40000, F - cnt - (F - cnt) % 16):
        n = (cnt - 1) * hop + N
        xs = night["xf"][f0 * hop: f0 * hop + n]
        torch.cuda.synchronize()
        e.stft_mag_f32_dev(xs.data_ptr(), 1, n, n, V, mags.data_ptr())
        sa._check(sa.lib().sono_detect_band_active_dev(e._handle(), mags.data_ptr(), cnt, N // 2, LO, HI, target, two.data_ptr(),
                                                       two_energy.data_ptr(), None))
        e.synchronize()
        assert torch.equal(two, flags[f0:f0 + cnt]), f0
        assert torch.allclose(two_energy, energy[f0:f0 + cnt], rtol=1e-12, atol=0), f0          # same terms, summed in another order
    # runs of active frames over the whole hour, both entry points, against the flags' own edges
    a = flags.to(torch.int8)
    prev = torch.cat([torch.zeros(1, dtype=torch.int8, device="cuda"), a[:-1]])
    nxt = torch.cat([a[1:], torch.zeros(1, dtype=torch.int8, device="cuda")])
    want_first = torch.nonzero((a == 1) & (prev == 0)).flatten()
    want_last = torch.nonzero((a == 1) & (nxt == 0)).flatten()
    for name in ("sono_detect_runs_dev", "sono_detect_runs_static_dev"):
        first = torch.zeros(F // 2 + 1, dtype=torch.int64, device="cuda")
        last = torch.zeros(F // 2 + 1, dtype=torch.int64, device="cuda")
        count = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        sa._check(getattr(sa.lib(), name)(e._handle(), flags.data_ptr(), F, first.data_ptr(), last.data_ptr(), count.data_ptr(), None))
        e.synchronize()
        k = int(count.item())
        assert k == want_first.numel() > 1000, (name, k)
        assert torch.equal(first[:k], want_first) and torch.equal(last[:k], want_last), name


def test_many_files_in_one_launch_equal_one_launch_per_file():
    """The batch shape of a folder of short recordings (thousands of 5 s files at 500 kHz dealt to one GPU, ONE launch
    over all of them): 1 200 files here, 12 GB of samples, 11.7 M frames.  Every row, peak and the global maximum of
    spot files equal, bit for bit, the same file transformed by a launch of its own (tiles never straddle files), no
    frame reaches across a file boundary (the count is files x frames per file), and the global maximum is the
    largest frame peak."""
    import torch
    sr, files, per = 500000, 1200, 5 * 500000
    V, H = 768, 256
    F1 = sa.frames_for(per, N, V)
    x = synth.chirp_torch(0, files * per, sr, device="cuda")
    e = sa.SpectrogramEngine(N, "hann")
    out = torch.empty(files * F1 * 512, dtype=torch.uint8, device="cuda")
    pb = torch.empty(files * F1, dtype=torch.int16, device="cuda")
    pm = torch.empty(files * F1, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e.stft_u8_dev(x.data_ptr(), files, per, per, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(), gm.data_ptr())
    e.synchronize()
    assert F1 == (per - N) // H + 1 and out.numel() == files * F1 * 512
    assert gm.view(torch.float32).item() == pm.max().item() > 0
    one = torch.empty(F1 * 512, dtype=torch.uint8, device="cuda")
    one_pb = torch.empty(F1, dtype=torch.int16, device="cuda")
    one_pm = torch.empty(F1, dtype=torch.float32, device="cuda")
    for f in (0, 1, 599, 1198, 1199):
        xs = x[f * per:(f + 1) * per]
        torch.cuda.synchronize()
        e.stft_u8_dev(xs.data_ptr(), 1, per, per, V, 20.0, 80.0, one.data_ptr(), one_pb.data_ptr(), one_pm.data_ptr())
        e.synchronize()
        assert torch.equal(one, out[f * F1 * 512:(f + 1) * F1 * 512]), f
        assert torch.equal(one_pb, pb[f * F1:(f + 1) * F1]) and torch.equal(one_pm, pm[f * F1:(f + 1) * F1]), f
    e.free()


def test_survey_pass_in_one_read_equals_the_two_passes(night):
    """sono_stft_survey(_pcm16)_dev: the peak features at hop N/4 and the fast scan's flags at hop N/2 from ONE launch over
    the hour (the scan's frames are the even frames of the feature pass).  Peak bins, peak magnitudes and the global
    maximum equal the feature pass bit for bit; the flags equal the fused fast scan's; the band sums agree to 1e-6 (a
    scan frame is transformed beside another partner frame here than in the hop-N/2 kernel, which reaches its bins at
    the level of the last bit of the f32 magnitudes, and the terms are added in another order).  Float and 16-bit forms, and a three-stream batch whose streams end on
    odd and even frame counts (the scan has ceil(frames / 2) frames per stream)."""
    torch, e = night["torch"], night["e"]
    V, VS = 768, 512
    F, FS = sa.frames_for(SAMPLES, N, V), sa.frames_for(SAMPLES, N, VS)
    assert FS == (F + 1) // 2
    target = (10.0 ** (-60.0 / 10.0)) * N * 0.25

    def outputs(frames, scan):
        return (torch.zeros(frames, dtype=torch.int16, device="cuda"), torch.zeros(frames, dtype=torch.float32, device="cuda"),
                torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(scan, dtype=torch.uint8, device="cuda"),
                torch.zeros(scan, dtype=torch.float64, device="cuda"))
    ref = outputs(F, FS)
    torch.cuda.synchronize()
    e.stft_features_dev(night["xf"].data_ptr(), 1, SAMPLES, SAMPLES, V, ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr())
    e.stft_band_active_dev(night["xf"].data_ptr(), 1, SAMPLES, SAMPLES, VS, LO, HI, target, ref[3].data_ptr(), ref[4].data_ptr())
    e.synchronize()
    assert 0.01 * FS < int(ref[3].sum().item()) < 0.5 * FS
    for pcm in (False, True):
        got = outputs(F, FS)
        torch.cuda.synchronize()
        if pcm:
            e.stft_survey_pcm16_dev(night["pcm"].data_ptr(), 1, SAMPLES, SAMPLES, 1, 0, LO, HI, target, *(t.data_ptr() for t in got))
        else:
            e.stft_survey_dev(night["xf"].data_ptr(), 1, SAMPLES, SAMPLES, LO, HI, target, *(t.data_ptr() for t in got))
        e.synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), pcm
        assert torch.equal(got[3], ref[3]), pcm
        assert torch.allclose(got[4], ref[4], rtol=1e-6, atol=0), pcm
    # three streams of 2 501 / 2 502 / 2 502 ... samples apart: 6, 7 frames ... per stream
    for per in (N + 5 * 256 + 3, N + 6 * 256, N + 7 * 256 + 255):
        f1, s1 = sa.frames_for(per, N, V), sa.frames_for(per, N, VS)
        assert s1 == (f1 + 1) // 2
        stride = per + 40
        x3 = night["xf"][384000: 384000 + 3 * stride]
        ref = outputs(3 * f1, 3 * s1)
        got = outputs(3 * f1, 3 * s1)
        torch.cuda.synchronize()
        e.stft_features_dev(x3.data_ptr(), 3, stride, per, V, ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr())
        e.stft_band_active_dev(x3.data_ptr(), 3, stride, per, VS, LO, HI, target, ref[3].data_ptr(), ref[4].data_ptr())
        e.stft_survey_dev(x3.data_ptr(), 3, stride, per, LO, HI, target, *(t.data_ptr() for t in got))
        e.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got[:4], ref[:4])), per
        assert torch.allclose(got[4], ref[4], rtol=1e-6, atol=0), per
    # another fft size takes the two passes one after the other behind the same entry point
    e2 = sa.SpectrogramEngine(512, "hann")
    n2 = 512 + 128 * 4001
    f2, s2 = sa.frames_for(n2, 512, 384), sa.frames_for(n2, 512, 256)
    ref, got = outputs(f2, s2), outputs(f2, s2)
    torch.cuda.synchronize()
    e2.stft_features_dev(night["xf"].data_ptr(), 1, n2, n2, 384, ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr())
    e2.stft_band_active_dev(night["xf"].data_ptr(), 1, n2, n2, 256, 13, 170, target, ref[3].data_ptr(), ref[4].data_ptr())
    e2.stft_survey_dev(night["xf"].data_ptr(), 1, n2, n2, 13, 170, target, *(t.data_ptr() for t in got))
    e2.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, ref))
    e2.free()
