"""The benchmark configurations at their FULL sizes, through size-independent checks (the oracle only sees spot
frames): the 10-minute high-resolution zoom (fft 4096, Blackman, 95 % overlap, 1 123 883 frames), a 12-hour 16-bit
channel (16.6 G samples: sample and byte offsets beyond 2^32 inside ONE stream) and 10 000 five-second files in one
launch (100 GB of samples, 50 GB of rows).  Frame-count rule: lib.rs:118-123."""
import numpy as np
import pytest

import spectrogram_amd as sa
from oracle import sono_oracle as so
from spectrogram_amd import synth
from tests import parity

pytestmark = pytest.mark.gpu

SR = 384000


def test_high_resolution_zoom_at_its_full_size():
    """fft 4096, Blackman (alpha 0.16), noverlap floor(4096 * 0.95) = 3891 (hop 205), 10 min at 384 kHz: 1 123 883
    frames in one launch.  Spot frames at the start, around the middle and at the very end against the C oracle; the
    recording cut into three chunks with an fft - hop halo gives the same rows, peaks and maxima bit for bit (chunk
    starts on tile boundaries, so every frame meets the partner it has in the one-launch form); the global maximum is
    the largest frame peak."""
    import torch
    N, V = 4096, 3891
    H, B = N - V, N // 2
    n = 600 * SR
    F = sa.frames_for(n, N, V)
    assert F == 1123883
    x = synth.chirp_torch(0, n, SR, stream=4, device="cuda")
    e = sa.SpectrogramEngine(N, "blackman", 0.16)
    out = torch.empty(F * B, dtype=torch.uint8, device="cuda")
    pb = torch.empty(F, dtype=torch.int16, device="cuda")
    pm = torch.empty(F, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e.stft_u8_dev(x.data_ptr(), 1, n, n, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(), gm.data_ptr())
    e.synchronize()
    assert gm.view(torch.float32).item() == pm.max().item() > 0
    # spot frames against the oracle (a pulse starts every 100 ms = 187.3 frames: the ranges see pulses and quiet frames)
    o = so.OracleEngine(N, "blackman", 0.16)
    nd = nt = 0
    cnt = 24
    for f0 in (0, 186, F // 2 - 7, F - cnt):
        xs = x[f0 * H: f0 * H + (cnt - 1) * H + N].cpu().numpy()
        ref = np.asarray(o.compute_spectrogram_u8(xs, V, 20.0, 80.0))
        got = out[f0 * B:(f0 + cnt) * B].cpu().numpy()
        a, b = parity.u8_diff(got, ref, f"zoom frames {f0}..")
        nd += a
        nt += b
        ref_pk, ref_pm = np.asarray(o.get_peaks(0.0)), np.asarray(o.get_peak_magnitudes(0.0))
        got_pk = pb[f0:f0 + cnt].cpu().numpy().view(np.uint16)
        got_pm = pm[f0:f0 + cnt].cpu().numpy()
        mags = np.asarray(o.last_magnitudes()).reshape(cnt, B)
        for i in np.flatnonzero(got_pk != ref_pk):          # only where the oracle's own magnitudes tie to 1e-6
            assert abs(mags[i, got_pk[i]] - mags[i, ref_pk[i]]) <= 1e-6 * mags[i].max(), (f0, i)
        assert np.all(np.abs(got_pm - ref_pm) <= 1e-5 * np.maximum(ref_pm, 1e-30)), f0
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    # three chunks with a halo, cut on tile boundaries (two frames per wave at this size)
    cuts = [0, (F // 3) // 2 * 2, (2 * F // 3) // 2 * 2, F]
    gmax_parts = []
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        nf = c1 - c0
        ns = (nf - 1) * H + N
        o_c = torch.empty(nf * B, dtype=torch.uint8, device="cuda")
        pb_c = torch.empty(nf, dtype=torch.int16, device="cuda")
        pm_c = torch.empty(nf, dtype=torch.float32, device="cuda")
        gm_c = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        e.stft_u8_dev(x.data_ptr() + 4 * c0 * H, 1, ns, ns, V, 20.0, 80.0, o_c.data_ptr(), pb_c.data_ptr(), pm_c.data_ptr(), gm_c.data_ptr())
        e.synchronize()
        assert torch.equal(o_c, out[c0 * B:c1 * B]), c0
        assert torch.equal(pb_c, pb[c0:c1]) and torch.equal(pm_c, pm[c0:c1]), c0
        gmax_parts.append(gm_c.view(torch.float32).item())
    assert max(gmax_parts) == gm.view(torch.float32).item()
    e.free()


def test_twelve_hour_pcm16_channel_in_one_stream():
    """12 h of one 384 kHz channel as 33 GB of int16 in HBM: 16 588 800 000 samples in ONE stream, so sample offsets
    (and the byte offsets of the samples) pass 2^32 a third of the way in.  The peak features at 75 % overlap and the
    fused fast scan at 50 % overlap run over the whole channel in one launch each; spot ranges -- the first frames,
    ranges on either side of sample 2^32, the last frames -- equal the float forms on the same samples converted
    (value / 32768), bit for bit; the run list equals the flags' own edges; the maximum is the largest frame peak.
    Hours 2 .. 12 are hour 1 with another constant added per hour: a wrapped offset lands in an hour whose samples differ."""
    import torch
    N, LO, HI = 1024, 26, 341
    hour = 3600 * SR
    n = 12 * hour
    assert n > 3 * (1 << 32)
    x = synth.chirp_torch(0, hour, SR, stream=9, device="cuda")
    first = torch.clamp(torch.round(x * 30000.0), -32768, 32767).to(torch.int16)
    del x
    pcm = torch.empty(n, dtype=torch.int16, device="cuda")
    for k in range(12):
        torch.add(first, 37 * k, out=pcm[k * hour:(k + 1) * hour])
    del first
    e = sa.SpectrogramEngine(N, "hann")

    def floats(s0, count):
        return pcm[s0:s0 + count].to(torch.float32) / 32768.0

    # ---- peak features, hop 256
    V, H = 768, 256
    F = sa.frames_for(n, N, V)
    assert F == 64799997
    pb = torch.empty(F, dtype=torch.int16, device="cuda")
    pm = torch.empty(F, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e.stft_u8_pcm16_dev(pcm.data_ptr(), 1, n, n, 1, 0, V, 20.0, 80.0, 0, pb.data_ptr(), pm.data_ptr(), gm.data_ptr())
    e.synchronize()
    assert gm.view(torch.float32).item() == pm.max().item() > 0
    cnt = 40000
    wrap = (1 << 32) // H                                   # the frame that starts at sample 2^32
    spots = [0, (wrap - cnt // 2) // 16 * 16, (2 * wrap + 123456) // 16 * 16, (3 * wrap - 16) // 16 * 16, (F - cnt) // 16 * 16]
    for f0 in spots:
        nf = min(cnt, F - f0)
        ns = (nf - 1) * H + N
        xs = floats(f0 * H, ns)
        pb_s = torch.empty(nf, dtype=torch.int16, device="cuda")
        pm_s = torch.empty(nf, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        e.stft_features_dev(xs.data_ptr(), 1, ns, ns, V, pb_s.data_ptr(), pm_s.data_ptr(), 0)
        e.synchronize()
        assert torch.equal(pb_s, pb[f0:f0 + nf]) and torch.equal(pm_s, pm[f0:f0 + nf]), f0
    del pb, pm

    # ---- fused fast scan, hop 512, and its run list
    V, H = 512, 512
    F = sa.frames_for(n, N, V)
    flags = torch.empty(F, dtype=torch.uint8, device="cuda")
    target = (10.0 ** (-60.0 / 10.0)) * N * 0.25            # batCallDetector.js:906-927 at -60 dB
    torch.cuda.synchronize()
    e.stft_band_active_pcm16_dev(pcm.data_ptr(), 1, n, n, 1, 0, V, LO, HI, target, flags.data_ptr(), 0)
    e.synchronize()
    active = int(flags.sum().item())
    assert 0.01 * F < active < 0.5 * F, active / F
    wrap = (1 << 32) // H
    for f0 in [0, (wrap - cnt // 2) // 16 * 16, (3 * wrap - 16) // 16 * 16, (F - cnt) // 16 * 16]:
        nf = min(cnt, F - f0)
        ns = (nf - 1) * H + N
        xs = floats(f0 * H, ns)
        fl = torch.empty(nf, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        e.stft_band_active_dev(xs.data_ptr(), 1, ns, ns, V, LO, HI, target, fl.data_ptr(), 0)
        e.synchronize()
        assert torch.equal(fl, flags[f0:f0 + nf]), f0
    a = flags.to(torch.int8)
    zero = torch.zeros(1, dtype=torch.int8, device="cuda")
    want_first = torch.nonzero((a == 1) & (torch.cat([zero, a[:-1]]) == 0)).flatten()
    want_last = torch.nonzero((a == 1) & (torch.cat([a[1:], zero]) == 0)).flatten()
    first_i = torch.zeros(F // 2 + 1, dtype=torch.int64, device="cuda")
    last_i = torch.zeros(F // 2 + 1, dtype=torch.int64, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    sa._check(sa.lib().sono_detect_runs_dev(e._handle(), flags.data_ptr(), F, first_i.data_ptr(), last_i.data_ptr(), count.data_ptr(), None))
    e.synchronize()
    k = int(count.item())
    assert k == want_first.numel() > 100000
    assert torch.equal(first_i[:k], want_first) and torch.equal(last_i[:k], want_last)
    e.free()
    del pcm, flags, a
    torch.cuda.empty_cache()


def test_ten_thousand_files_in_one_launch():
    """10 000 five-second 500 kHz files (2 500 000 samples, 9 762 frames each) in ONE launch: 100 GB of samples, 97.62 M
    frames, 50 GB of rows -- sample, frame and byte offsets all far beyond 2^32.  Spot files, the first, the last and
    those on either side of the offsets' 2^32 marks among them, equal the same file transformed by a launch of its
    own, bit for bit (tiles never straddle files); the global maximum is the largest frame peak.  The files are one
    hundred distinct synthetic files, repeated with another gain per repetition."""
    import torch
    N, V, H, B = 1024, 768, 256, 512
    sr, files, per = 500000, 10000, 2500000
    F1 = sa.frames_for(per, N, V)
    assert F1 == 9762
    base = synth.chirp_torch(0, 100 * per, sr, stream=2, device="cuda")
    x = torch.empty(files * per, dtype=torch.float32, device="cuda")
    for k in range(100):
        torch.mul(base, 1.0 - 0.004 * k, out=x[k * 100 * per:(k + 1) * 100 * per])
    del base
    e = sa.SpectrogramEngine(N, "hann")
    out = torch.empty(files * F1 * B, dtype=torch.uint8, device="cuda")
    pb = torch.empty(files * F1, dtype=torch.int16, device="cuda")
    pm = torch.empty(files * F1, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e.stft_u8_dev(x.data_ptr(), files, per, per, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(), gm.data_ptr())
    e.synchronize()
    assert gm.view(torch.float32).item() == pm.max().item() > 0
    one = torch.empty(F1 * B, dtype=torch.uint8, device="cuda")
    one_pb = torch.empty(F1, dtype=torch.int16, device="cuda")
    one_pm = torch.empty(F1, dtype=torch.float32, device="cuda")
    sample_wrap = (1 << 32) // per                       # the file that holds sample 2^32
    byte_wrap = (1 << 32) // (F1 * B)                    # the file whose rows hold output byte 2^32
    spots = sorted({0, 1, byte_wrap, byte_wrap + 1, sample_wrap, sample_wrap + 1, 4999, 2 * sample_wrap + 1, files - 2, files - 1})
    for f in spots:
        xs = x[f * per:(f + 1) * per]
        torch.cuda.synchronize()
        e.stft_u8_dev(xs.data_ptr(), 1, per, per, V, 20.0, 80.0, one.data_ptr(), one_pb.data_ptr(), one_pm.data_ptr())
        e.synchronize()
        assert torch.equal(one, out[f * F1 * B:(f + 1) * F1 * B]), f
        assert torch.equal(one_pb, pb[f * F1:(f + 1) * F1]) and torch.equal(one_pm, pm[f * F1:(f + 1) * F1]), f
    e.free()
    del x, out, pb, pm
    torch.cuda.empty_cache()


def test_whole_file_detection_at_production_piece_size():
    """Forty minutes of one 16-bit channel through sono_process_full_file_pcm16 as a survey runs it: the recording
    crosses the link in pieces of 2^26 samples (fourteen of them here) and its 24 000 regions are handed on in eight
    groups while later pieces are still on their way.  Every row of the call table equals the one-piece form's bit for
    bit; the pulses come every 100 ms, so there are as many calls as tenths of a second."""
    import torch
    from spectrogram_amd import framing as frm
    seconds = 2400
    n = seconds * SR
    x = synth.chirp_torch(0, n, SR, stream=0, device="cuda")
    q = torch.clamp(torch.round(x * 32767.0), -32768, 32767).to(torch.int16).cpu().numpy()
    del x
    torch.cuda.empty_cache()
    e = sa.SpectrogramEngine(1024, "hann")
    tables = []
    for whole in (False, True):
        e.set_option("no_upload_overlap", whole)
        tables.append(frm.process_full_file_table(e, q, SR, 10.0, 128.0, None, {"debugMode": True}, file_id=7))
    piecewise, one_piece = tables
    assert piecewise.shape == one_piece.shape and abs(piecewise.shape[0] - seconds * 10) <= 2
    assert np.array_equal(piecewise.view(np.uint64), one_piece.view(np.uint64))
    starts = piecewise[:, 3]                                  # start times rise with the regions
    assert np.all(np.diff(starts) > 0)
    e.free()
