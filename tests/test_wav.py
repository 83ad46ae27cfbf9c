"""WAV / PCM ingest: the RIFF chunk walk on any machine, the GPU conversion against numpy.
The reference decodes in the browser (WebAudio), so there is no reference vector for the sample
scaling: integer PCM / 2^(bits-1) is the WebAudio convention -- parity unpinned."""
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
    per = N + 29 * H + 3                                     # sample frames per stream; the last tile is partial
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
    per = N + 37 * H + 5
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
