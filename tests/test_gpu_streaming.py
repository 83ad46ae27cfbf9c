"""Long host-pointer calls of compute_spectrogram_u8 go through the chunked pipeline (pinned rings, upload /
kernel / download overlapped): rows, peaks and global maximum must equal the one-copy-each-way form bit for bit."""
import os

import numpy as np
import pytest

import spectrogram_amd as sa
from spectrogram_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_fft,noverlap,chunks", [(1024, 768, 4.3), (512, 256, 3.1), (1024, 700, 2.2), (4096, 3891, 2.6)])
def test_streamed_rows_equal_one_copy_rows(n_fft, noverlap, chunks, monkeypatch):
    hop = n_fft - noverlap
    per_chunk = max((32 << 20) // (4 * hop), 4096) // 16 * 16          # stream_chunk_frames
    frames = int(per_chunk * chunks) + 7
    n = (frames - 1) * hop + n_fft + 3
    x = synth.chirp_numpy(0, n, 384000, stream=5)
    x[n // 3: n // 3 + 4 * n_fft] = 0.0                                 # some exactly silent frames
    e = sa.SpectrogramEngine(n_fft, "blackman" if n_fft == 4096 else "hann")
    e.set_option("no_streaming", False)
    a = e.compute_spectrogram_u8(x, noverlap, 20.0, 80.0)
    pa, ma, ga = e.get_peaks(0.3), e.get_peak_magnitudes(0.0), e.get_global_max()
    e.set_option("no_streaming", True)
    b = e.compute_spectrogram_u8(x, noverlap, 20.0, 80.0)
    pb, mb, gb = e.get_peaks(0.3), e.get_peak_magnitudes(0.0), e.get_global_max()
    assert a.shape == b.shape == (frames * (n_fft // 2),)
    assert np.array_equal(a, b) and np.array_equal(pa, pb) and np.array_equal(ma, mb) and ga == gb
    # the rings are reused by a second, shorter streamed call
    e.set_option("no_streaming", False)
    m = (int(per_chunk * 2.05) - 1) * hop + n_fft
    c = e.compute_spectrogram_u8(x[:m], noverlap, 20.0, 80.0)
    assert np.array_equal(c, b[: c.size])
    e.free()
