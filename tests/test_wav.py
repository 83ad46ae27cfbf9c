"""WAV / PCM ingest: the RIFF chunk walk on any machine, the GPU conversion against numpy.
The reference decodes in the browser (WebAudio), so there is no reference vector for the sample
scaling: integer PCM / 2^(bits-1) is the WebAudio convention -- parity unpinned.  The header walk is
checked against hand-derived cases (expected values from reading fileLoader.js:7-63) and a CPU
restatement of its two functions (oracle/wav_oracle.py): unpinned as well, nothing of the reference
was run for them."""
import struct

import numpy as np
import pytest

import spectrogram_amd as sa


def make_wav(samples, sr, bits, channels=1, is_float=False, extra_chunks=(), odd_pad_chunk=True):
    """samples: int array (frames x channels) already in the sample type's range, or float32."""
    a = np.asarray(samples).reshape(-1, channels)
    if is_float:
        data = a.astype("<f4").tobytes()
    elif bits == 8:
        data = a.astype(np.uint8).tobytes()
    elif bits == 16:
        data = a.astype("<i2").tobytes()
    elif bits == 24:
        v = a.astype("<i4").reshape(-1)
        data = b"".join(struct.pack("<i", int(x))[:3] for x in v)
    else:
        data = a.astype("<i4").tobytes()
    fmt = struct.pack("<HHIIHH", 3 if is_float else 1, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits)
    chunks = [(b"fmt ", fmt)]
    if odd_pad_chunk:
        chunks.append((b"LIST", b"abc"))                 # odd size: the walk must skip the pad byte (fileLoader.js:23)
    chunks += list(extra_chunks) + [(b"data", data)]
    body = b"WAVE"
    for cid, payload in chunks:
        body += cid + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) % 2 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_parse_walks_chunks_like_the_reference():
    x = (np.arange(10) * 1000 - 5000).astype(np.int16)
    w = make_wav(np.stack([x, -x], axis=1), 384000, 16, channels=2, extra_chunks=[(b"guan", b"GUANO|Version:1.0\n")])
    info = sa.wav_info(w)
    assert info["sample_rate"] == 384000 and info["channels"] == 2 and info["bits_per_sample"] == 16
    assert info["frames"] == 10 and info["format_tag"] == 1 and info["data_bytes"] == 40
    assert w[info["data_offset"]:info["data_offset"] + 4] == struct.pack("<hh", -5000, 5000)
    assert w[info["guano_offset"]:info["guano_offset"] + info["guano_bytes"]] == b"GUANO|Version:1.0\n"
    # a data chunk that claims more bytes than the file holds is clipped
    cut = sa.wav_info(w[:-8])
    assert cut["frames"] == 8
    for bad in (b"", b"RIFF\0\0\0\0WAVX", w[:20]):
        with pytest.raises(sa.SonoError):
            sa.wav_info(bad)


def riff(chunks, pad=True):
    body = b"WAVE"
    for cid, payload in chunks:
        body += cid + struct.pack("<I", len(payload)) + payload + (b"\0" if pad and len(payload) % 2 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def fmt_chunk(sr, channels=1, bits=16, tag=1, extra=b""):
    return struct.pack("<HHIIHH", tag, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits) + extra


# Files whose expected values follow from reading modules/fileLoader.js:7-63 (nothing of the reference was run:
# parity unpinned); oracle/wav_oracle.py restates the two functions and is checked against the same literals.
HAND_CASES = [
    # name, file, sample rate (getWavSampleRate), channels, bits, declared data size, duration or None (walk raises)
    ("plain", riff([(b"fmt ", fmt_chunk(384000)), (b"data", bytes(200))]), 384000, 1, 16, 200, 100 / 384000),
    ("odd chunks before fmt and data",
     riff([(b"junk", b"abc"), (b"fmt ", fmt_chunk(500000, 2, 16)), (b"LIST", b"x"), (b"data", bytes(400))]),
     500000, 2, 16, 400, 100 / 500000),
    # two 'fmt ' chunks in front of 'data': getWavSampleRate returns at the first (js:20-22), getWavDuration keeps the last (js:46-49)
    ("two fmt chunks", riff([(b"fmt ", fmt_chunk(96000)), (b"fmt ", fmt_chunk(192000, 2, 24)), (b"data", bytes(600))]),
     96000, 2, 24, 600, 100 / 192000),
    # a 'fmt ' chunk behind 'data' is never reached by getWavDuration (break, js:52)
    ("fmt after data", riff([(b"fmt ", fmt_chunk(256000)), (b"data", bytes(64)), (b"fmt ", fmt_chunk(8000, 2, 8))]),
     256000, 1, 16, 64, 32 / 256000),
    # a data chunk that declares more than the file holds: the declared size counts (js:51, 57-60)
    ("truncated data", riff([(b"fmt ", fmt_chunk(384000)), (b"data", bytes(100))])[:-40], 384000, 1, 16, 100, 50 / 384000),
    ("extensible format", riff([(b"fmt ", fmt_chunk(250000, 1, 16, 0xFFFE, struct.pack("<HHI", 22, 16, 4) + struct.pack("<H", 1) + bytes(14))),
                                (b"data", bytes(50))]), 250000, 1, 16, 50, 25 / 250000),
    # header shapes of real recorders: a Broadcast Wave 'bext' chunk (602 bytes of description, originator, time reference,
    # UMID ... plus an odd-length coding history, so a pad byte follows) in front of 'fmt '; a 'LIST'/'INFO' chunk in front
    # of 'fmt '; a 'fact' chunk between 'fmt ' and 'data' (float and extensible files carry one)
    ("bext before fmt", riff([(b"bext", bytes(602) + b"A=PCM,F=384000,W=16,M=mono,T=recorder\r\n"), (b"fmt ", fmt_chunk(384000)),
                              (b"data", bytes(120))]), 384000, 1, 16, 120, 60 / 384000),
    ("LIST INFO before fmt", riff([(b"LIST", b"INFOINAM" + struct.pack("<I", 5) + b"bat1\0\0"), (b"fmt ", fmt_chunk(300000, 2, 16)),
                                   (b"data", bytes(80))]), 300000, 2, 16, 80, 20 / 300000),
    ("fact between fmt and data", riff([(b"fmt ", fmt_chunk(192000, 1, 32, 3)), (b"fact", struct.pack("<I", 25)), (b"data", bytes(100))]),
     192000, 1, 32, 100, 25 / 192000),
    # no 'fmt ': the rate falls back to 256000 (js:26) and the duration to 0 (js:57, 62)
    ("no fmt", riff([(b"LIST", b"abcd"), (b"data", bytes(32))]), 256000, 1, 16, 32, 0.0),
    ("no data", riff([(b"fmt ", fmt_chunk(384000))]), 384000, 1, 16, 0, 0.0),
    # the header of the last chunk ends exactly at the end of the file: `pos < byteLength - 8` is false, it is not looked at
    ("data header at the very end", riff([(b"fmt ", fmt_chunk(384000))]) + b"data" + struct.pack("<I", 16), 384000, 1, 16, 0, 0.0),
]


@pytest.mark.parametrize("case", HAND_CASES, ids=[c[0] for c in HAND_CASES])
def test_wav_header_hand_derived_cases(case):
    from oracle import wav_oracle as wo
    name, w, rate, channels, bits, data_size, duration = case
    assert wo.get_wav_sample_rate(w) == rate
    parts = wo.get_wav_duration_parts(w)
    assert (parts["channels"], parts["bits_per_sample"], parts["data_size"]) == (channels, bits, data_size)
    assert parts["duration"] == pytest.approx(duration, rel=1e-12, abs=0.0)
    # sono_wav_parse against the same reading.  Where the two reference functions disagree (two 'fmt ' chunks) it
    # keeps the first chunk, as getWavSampleRate does; it clips a data chunk to the file; and it reports a file
    # without fmt or data, which the reference answers with its defaults, as an error.
    if name in ("no fmt", "no data", "data header at the very end"):
        with pytest.raises(sa.SonoError):
            sa.wav_info(w)
        return
    info = sa.wav_info(w)
    assert info["sample_rate"] == rate
    if name == "two fmt chunks":
        assert (info["channels"], info["bits_per_sample"]) == (1, 16)
        return
    assert (info["channels"], info["bits_per_sample"]) == (channels, bits)
    assert info["data_offset"] == parts["data_pos"]
    held = min(data_size, len(w) - parts["data_pos"])
    assert info["data_bytes"] == held and info["frames"] == held // (channels * bits // 8)
    if name != "truncated data":
        assert info["frames"] / info["sample_rate"] == pytest.approx(duration, rel=1e-12)
    if name == "extensible format":
        assert info["format_tag"] == 1                       # the sub-format of the extension


@pytest.mark.gpu
@pytest.mark.parametrize("bits,is_float", [(8, False), (16, False), (24, False), (32, False), (32, True)])
def test_decode_wav_channel(bits, is_float):
    rng = np.random.default_rng(bits)
    n, ch = 5001, 3
    if is_float:
        a = rng.uniform(-1, 1, (n, ch)).astype(np.float32)
        ref = a
    elif bits == 8:
        a = rng.integers(0, 256, (n, ch))
        ref = ((a.astype(np.float32) - 128) / 128).astype(np.float32)
    else:
        lim = 1 << (bits - 1)
        a = rng.integers(-lim, lim, (n, ch))
        a[0, :] = [-lim, lim - 1, 0]
        ref = (a.astype(np.float64) / lim).astype(np.float32)
    w = make_wav(a, 500000, bits, channels=ch, is_float=is_float)
    e = sa.SpectrogramEngine(256, "hann")
    for c in range(ch):
        got, info = sa.decode_wav(e, w, c)
        assert info["frames"] == n and got.dtype == np.float32 and np.array_equal(got, ref[:, c])
    with pytest.raises(sa.SonoError):
        sa.decode_wav(e, w, ch)
    e.free()


@pytest.mark.gpu
@pytest.mark.parametrize("frames,byte_offset,out_offset", [(4096, 0, 0), (40003, 0, 0), (40003, 4, 0), (40003, 2, 0), (40003, 0, 1),
                                                            (1234567, 44, 0)])
def test_pcm16_mono_conversion_paths(frames, byte_offset, out_offset):
    # mono 16-bit PCM in HBM: the eight-samples-per-thread path (data on a multiple of four bytes, output on 16), its tail,
    # and the general path for the other alignments -- all equal to value / 32768
    import torch
    rng = np.random.default_rng(frames + byte_offset)
    a = rng.integers(-32768, 32768, frames + 8, dtype=np.int64).astype(np.int16)
    a[:3] = [-32768, 32767, 0]
    raw = torch.from_numpy(np.frombuffer(b"\0" * byte_offset + a.tobytes(), np.uint8).copy()).cuda()
    out = torch.full((frames + out_offset + 4,), 7.0, dtype=torch.float32, device="cuda")
    e = sa.SpectrogramEngine(256, "hann")
    sa._check(sa.lib().sono_pcm_to_f32_dev(e._handle(), raw.data_ptr() + byte_offset, frames, 1, 0, 16, 0,
                                           out.data_ptr() + 4 * out_offset, None))
    e.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[out_offset:out_offset + frames], a[:frames].astype(np.float32) / 32768)
    assert (got[:out_offset] == 7.0).all() and (got[out_offset + frames:] == 7.0).all()         # nothing written outside
    e.free()


@pytest.mark.gpu
def test_wav_to_u8_frames_equals_float_input():
    # int16 WAV -> decode -> compute_spectrogram_u8 equals feeding the same samples as floats
    rng = np.random.default_rng(1)
    t = np.arange(6000)
    x16 = np.round(8000 * np.sin(0.3 * t + 2e-5 * t * t) + 20 * rng.standard_normal(t.size)).astype(np.int16)
    w = make_wav(x16, 384000, 16)
    e = sa.SpectrogramEngine(512, "hann")
    x, info = sa.decode_wav(e, w)
    assert np.array_equal(x, x16.astype(np.float32) / 32768)
    assert np.array_equal(e.compute_spectrogram_u8(x, 256, 20.0, 80.0),
                          e.compute_spectrogram_u8(x16.astype(np.float32) / 32768, 256, 20.0, 80.0))
    e.free()


@pytest.mark.gpu
@pytest.mark.parametrize("N,V,channels,channel", [(1024, 768, 1, 0), (1024, 700, 2, 1), (256, 192, 2, 0), (512, 256, 1, 0),
                                                  (2048, 1536, 3, 2), (4096, 3891, 1, 0)])
def test_pcm16_frames_in_hbm_equal_convert_then_stft(N, V, channels, channel):
    # the STFT kernel reading the WAV data chunk's int16 sample frames itself (no f32 copy of the recording)
    # against sono_pcm_to_f32_dev followed by sono_stft_u8_dev: every output bit for bit
    import torch
    H, B, S = N - V, N // 2, 3
    per = N + 28 * H + 3                                     # 29 frames per stream: the last tile is partial and the
                                                             # last frame has no partner in the kernel's frame pairs
    stride = per + 7
    rng = np.random.default_rng(N + channels)
    t = np.arange(S * stride)
    pcm = np.zeros((S * stride, channels), np.int16)
    for c in range(channels):
        pcm[:, c] = np.round(9000 * np.sin((0.2 + 0.05 * c) * t + 1e-5 * t * t) + 30 * rng.standard_normal(t.size)).astype(np.int16)
    pcm[: N + H, channel] = 0                                # all-zero frames
    pcm[N + 6 * H: N + 7 * H, channel] = 32767               # a burst only some frames see; full scale
    pcm[N + 9 * H, channel] = -32768
    d_pcm = torch.from_numpy(pcm.reshape(-1)).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    F = sa.frames_for(per, N, V)
    T = S * F

    def outputs():
        return (torch.full((T * B,), 77, dtype=torch.uint8, device="cuda"), torch.zeros(T, dtype=torch.int16, device="cuda"),
                torch.zeros(T, dtype=torch.float32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"))

    # reference composition: convert the whole buffer's channel, then the f32 kernel
    d_x = torch.zeros(S * stride, dtype=torch.float32, device="cuda")
    sa._check(sa.lib().sono_pcm_to_f32_dev(e._handle(), d_pcm.data_ptr(), S * stride, channels, channel, 16, 0, d_x.data_ptr(), None))
    want = outputs()
    e.stft_u8_dev(d_x.data_ptr(), S, stride, per, V, 20.0, 80.0, want[0].data_ptr(), want[1].data_ptr(), want[2].data_ptr(),
                  want[3].data_ptr())
    got = outputs()
    e.stft_u8_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channel, V, 20.0, 80.0, got[0].data_ptr(), got[1].data_ptr(),
                        got[2].data_ptr(), got[3].data_ptr())
    e.synchronize()
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    assert int(want[0].max()) > 0 and not bool(want[0][: 2 * B].any())
    # features only, and argument checks
    feat = outputs()
    e.stft_u8_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channel, V, 20.0, 80.0, 0, feat[1].data_ptr(), feat[2].data_ptr(),
                        feat[3].data_ptr())
    e.synchronize()
    assert torch.equal(feat[1], want[1]) and torch.equal(feat[2], want[2]) and torch.equal(feat[3], want[3])
    with pytest.raises(sa.SonoError):
        e.stft_u8_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channels, V, 20.0, 80.0, got[0].data_ptr())
    with pytest.raises(sa.SonoError):
        e.stft_u8_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channel, N, 20.0, 80.0, got[0].data_ptr())
    with pytest.raises(sa.SonoError):
        e.stft_u8_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channel, V, 20.0, 80.0)
    e.free()


@pytest.mark.gpu
@pytest.mark.parametrize("N,V,channels,channel", [(1024, 512, 1, 0), (1024, 600, 2, 1), (512, 256, 2, 0), (4096, 2048, 1, 0)])
def test_pcm16_fast_scan_equals_convert_then_scan(N, V, channels, channel):
    # the fused fast scan reading int16 sample frames against sono_pcm_to_f32_dev + sono_stft_band_active_dev
    import torch
    H, S = N - V, 2
    per = N + 36 * H + 5                                     # 37 frames: the last one has no partner in the kernel's pairs
    stride = per + 3
    rng = np.random.default_rng(7 * N + channels)
    t = np.arange(S * stride)
    pcm = np.zeros((S * stride, channels), np.int16)
    for c in range(channels):
        gate = (rng.random(t.size) > 0.4)
        pcm[:, c] = np.round(6000 * np.sin((0.3 + 0.04 * c) * t + 2e-5 * t * t) * gate + 40 * rng.standard_normal(t.size)).astype(np.int16)
    pcm[: N + 2 * H, channel] = 0
    d_pcm = torch.from_numpy(pcm.reshape(-1)).cuda()
    e = sa.SpectrogramEngine(N, "hann")
    F = sa.frames_for(per, N, V)
    T = S * F
    lo, hi = 5, N // 2 - 3
    d_x = torch.zeros(S * stride, dtype=torch.float32, device="cuda")
    sa._check(sa.lib().sono_pcm_to_f32_dev(e._handle(), d_pcm.data_ptr(), S * stride, channels, channel, 16, 0, d_x.data_ptr(), None))
    act = [torch.full((T,), 9, dtype=torch.uint8, device="cuda") for _ in range(2)]
    en = [torch.zeros(T, dtype=torch.float64, device="cuda") for _ in range(2)]
    e.stft_band_active_dev(d_x.data_ptr(), S, stride, per, V, lo, hi, 1.0, act[0].data_ptr(), en[0].data_ptr(), 0)
    e.synchronize()
    target = float(en[0].median().item())
    e.stft_band_active_dev(d_x.data_ptr(), S, stride, per, V, lo, hi, target, act[0].data_ptr(), en[0].data_ptr(), 0)
    e.stft_band_active_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channel, V, lo, hi, target, act[1].data_ptr(),
                                 en[1].data_ptr(), 0)
    e.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(en[0], en[1]) and torch.equal(act[0], act[1])
    assert 0 < int(act[1].sum()) < T and float(en[1][0]) == 0.0
    with pytest.raises(sa.SonoError):
        e.stft_band_active_pcm16_dev(d_pcm.data_ptr(), S, stride, per, channels, channel, V, lo, N // 2, target, act[1].data_ptr())
    e.free()
