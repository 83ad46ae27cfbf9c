"""TEST INFRASTRUCTURE ONLY -- numpy restatement of compute_spectrogram_image and its helper
compute_frame_spectrum (spectrogram-wasm/src/lib.rs:494-616, 618-675): per-frame u8 spectra with the
image path's own quantisation (NOT the one of compute_spectrogram_u8), bilinear resampling of the
frames x bins grid onto width x height pixels in f32, colour map lookup.  Nothing outside tests/ may
import this.  The reference holds no test or fixture for this export and has no caller for it, so
the restatement is "parity unpinned": it follows the Rust text line by line, on magnitudes that come
from the pinned C oracle."""
import numpy as np


def frame_spectrum_u8(mags, gain_db, range_db, bank=None, num_filters=0):
    """compute_frame_spectrum (lib.rs:618-675) for every frame: mags is frames x bins (f32, the linear
    magnitudes of lib.rs:634-638)."""
    m = np.asarray(mags, np.float32)
    if bank is not None and num_filters and len(bank):
        bins = m.shape[1]
        flen = bins + 1                                          # filter_len = fft_size / 2 + 1, lib.rs:644
        out = np.zeros((m.shape[0], num_filters), np.float32)
        for i in range(num_filters):
            row = np.asarray(bank[i * flen: i * flen + bins], np.float32)      # lib.rs:647-648
            acc = np.zeros(m.shape[0], np.float32)
            for j in range(len(row)):                            # sequential f32 sum, lib.rs:650-652
                acc = (acc + m[:, j] * row[j]).astype(np.float32)
            out[:, i] = acc
        m = out
    f32 = np.float32
    with np.errstate(divide="ignore", invalid="ignore"):
        db = np.where(m > 0, f32(20.0) * np.log10(m, dtype=np.float32), f32(-80.0)).astype(np.float32)        # lib.rs:661-665
    normalized = ((db + f32(range_db) / f32(2.0) + f32(gain_db)) / f32(range_db)).astype(np.float32)          # lib.rs:668
    clamped = np.clip(normalized, f32(0), f32(1))
    return (clamped * f32(255.0)).astype(np.float32).astype(np.uint8)                                        # `as u8` truncates


def spectrogram_image(mags, color_map_rgba, width, height, gain_db, range_db, bank=None, num_filters=0):
    """compute_spectrogram_image (lib.rs:494-616).  mags: frames x bins; color_map_rgba: 1024 bytes or None.
    -> uint8 array of width * height * 4."""
    out = np.zeros(width * height * 4, np.uint8)
    if width == 0 or height == 0 or color_map_rgba is None or len(color_map_rgba) != 1024:
        return out                                               # lib.rs:504-506, 452-454
    mags = np.asarray(mags, np.float32)
    frames = mags.shape[0]
    if frames == 0:
        return out                                               # lib.rs:523-525
    spec = frame_spectrum_u8(mags, gain_db, range_db, bank, num_filters)
    use_bank = bank is not None and num_filters > 0
    spec_height = num_filters if use_bank else mags.shape[1]     # lib.rs:528-532
    cols = spec.shape[1]
    f32 = np.float32
    time_step = f32(frames) / f32(width)                         # lib.rs:540-541
    freq_step = f32(spec_height) / f32(height)
    cmap = np.asarray(color_map_rgba, np.uint8).reshape(256, 4)
    px = out.reshape(height, width, 4)
    for y in range(height):
        sf = f32(height - 1 - y) * freq_step                     # lib.rs:546-552
        fi = int(np.floor(sf))
        ff = f32(sf - f32(fi))
        f0, f1 = min(fi, spec_height - 1), min(fi + 1, spec_height - 1)
        for x in range(width):
            st = f32(x) * time_step                              # lib.rs:556-563
            ti = int(np.floor(st))
            tf = f32(st - f32(ti))
            t0, t1 = min(ti, frames - 1), min(ti + 1, frames - 1)
            magnitude = f32(0)
            for t in (t0, t1):                                   # lib.rs:569-600
                for fq in (f0, f1):
                    if fq < cols:
                        val = f32(spec[t, fq]) / f32(255.0)
                        tw = f32(1.0) - tf if t == t0 else tf
                        fw = f32(1.0) - ff if fq == f0 else ff
                        magnitude = f32(magnitude + f32(f32(val * tw) * fw))
            idx = int(np.clip(f32(magnitude * f32(255.0)), 0.0, 255.0))                                      # lib.rs:603
            px[y, x] = cmap[idx]
    return out
