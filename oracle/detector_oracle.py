"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the detector front end of the reference
(/root/reference/modules/batCallDetector.js): the array passes that sit
directly on the STFT output.  Only tests/ may import it.  The STFT itself comes
from the C oracle (sono_oracle.OracleEngine).

Parity status: PINNED by tests/golden/golden_detector.npz, captured by
oracle/make_golden_detector.py from the reference's own JavaScript methods
running against the reference wasm.
"""
import math

import numpy as np


def fast_scan_segments(engine, audio, sample_rate, flow_khz, fhigh_khz, threshold_db):
    """fastScanSegmentsWasm, batCallDetector.js:878-949 -> list of (start, end) samples."""
    audio = np.ascontiguousarray(audio, np.float32)
    fft = engine.get_fft_size()
    hop = fft // 2                                           # :885
    freq_res = sample_rate / fft                             # :889
    min_bin = math.floor(flow_khz * 1000 / freq_res)         # :890
    max_bin = math.ceil(fhigh_khz * 1000 / freq_res)         # :891
    raw = np.asarray(engine.compute_spectrogram(audio, fft - hop), np.float64)   # :894
    bins = engine.get_freq_bins()
    frames = len(raw) // bins                                # :897
    target = math.pow(10, threshold_db / 10) * fft * 0.25    # :906-909
    segments, active_start = [], None
    for f in range(frames):
        lo = f * bins + min_bin
        hi = min(f * bins + max_bin, len(raw) - 1)           # flat index; `break` past the array end (:919)
        band = raw[lo:hi + 1] if hi >= lo else raw[:0]
        energy = 0.0
        for m in band:                                       # sequential f64 sum (:918-925); terms are >= 0, so the
            energy += m * m                                  # early exit at :924 does not change the comparison below
        is_active = energy > target                          # :927
        sample_index = f * hop
        if is_active:
            if active_start is None:
                active_start = sample_index
        elif active_start is not None:
            segments.append((active_start, sample_index + fft))   # :937
            active_start = None
    if active_start is not None:
        segments.append((active_start, len(audio)))          # :944-946
    return segments


def merge_and_pad_segments(segments, total_samples, sample_rate, padding_ms):
    """mergeAndPadSegments, batCallDetector.js:1028-1056 -> list of (startSample, endSample)."""
    if not segments:
        return []
    pad = math.floor((padding_ms / 1000) * sample_rate + 0.5)     # Math.round
    srt = sorted(segments, key=lambda s: s[0])
    merged = []
    cur_start = max(0, srt[0][0] - pad)
    cur_end = min(total_samples, srt[0][1] + pad)
    for s, e in srt[1:]:
        ns, ne = max(0, s - pad), min(total_samples, e + pad)
        if ns <= cur_end:
            cur_end = max(cur_end, ne)
        else:
            merged.append((cur_start, cur_end))
            cur_start, cur_end = ns, ne
    merged.append((cur_start, cur_end))
    return merged


def generate_spectrogram(engine, audio, sample_rate, flow_khz, fhigh_khz, hop_percent=3.125):
    """generateSpectrogramWasm, batCallDetector.js:1296-1377 -> dict or None."""
    audio = np.ascontiguousarray(audio, np.float32)
    fft = engine.get_fft_size()
    hop = math.floor(fft * (hop_percent / 100))              # :1306
    if hop < 1 or fft > len(audio):                          # :1309
        return None
    raw = np.asarray(engine.compute_spectrogram(audio, fft - hop), np.float64)
    bins = engine.get_freq_bins()
    freq_res = sample_rate / fft
    frames = len(raw) // bins
    if frames < 1 or bins < 1 or len(raw) == 0:              # :1322
        return None
    min_bin = max(0, math.floor(flow_khz * 1000 / freq_res)) # :1328
    max_bin = min(bins - 1, math.floor(fhigh_khz * 1000 / freq_res))
    nb = max_bin - min_bin + 1
    if nb <= 0:                                              # :1332
        return None
    freq_bins = ((min_bin + np.arange(nb)) * freq_res).astype(np.float32)        # :1342-1344
    mag = raw.reshape(frames, bins)[:, min_bin:max_bin + 1]
    psd = (mag * mag) / fft                                  # :1357-1359, doubles
    power = (10 * np.log10(np.maximum(psd, 1e-16))).astype(np.float32)           # :1360 into a Float32Array
    time_frames = (np.arange(frames) * hop + fft / 2) / sample_rate              # :1350-1351
    return dict(powerMatrix=power, timeFrames=time_frames, freqBins=freq_bins, freqResolution=freq_res,
                minBin=min_bin)


def detect_call_segments(power_matrix, call_threshold_db=-24.0):
    """detectCallSegments, batCallDetector.js:1540-1598 -> list of (startFrame, endFrame)."""
    p = np.asarray(power_matrix, np.float32)
    if p.shape[0] == 0:
        return []
    gmax = float(p.max()) if p.size else -math.inf           # :1544-1550
    threshold = gmax + call_threshold_db                     # :1553
    active = (p.astype(np.float64) > threshold).any(axis=1)  # :1556-1567
    return runs(active)


def runs(active):
    """Maximal runs of True -> list of (first, last) (batCallDetector.js:1570-1595)."""
    out, start = [], None
    for i, a in enumerate(active):
        if a:
            if start is None:
                start = i
        elif start is not None:
            out.append((start, i - 1))
            start = None
    if start is not None:
        out.append((start, len(active) - 1))
    return out


def frame_features(power_matrix, peak_b0, peak_b1, low_max_bin, low_threshold_db, high_min_bin, high_threshold_db):
    """Per-frame scans of measureFrequencyParameters: arg-max within a band
    (batCallDetector.js:3520-3527, 3612-3630), lowest bin above a threshold
    (:2509-2522), highest bin above a threshold (:2002-2038).
    -> (peak_bin, peak_db, low_bin, high_bin) arrays; -1 / -inf where nothing qualifies."""
    p = np.asarray(power_matrix, np.float32)
    F, nb = p.shape
    peak_bin = np.full(F, -1, np.int32)
    peak_db = np.full(F, -np.inf, np.float32)
    low_bin = np.full(F, -1, np.int32)
    high_bin = np.full(F, -1, np.int32)
    for f in range(F):
        row = p[f]
        best, idx = -math.inf, -1
        for b in range(peak_b0, min(peak_b1, nb - 1) + 1):
            if row[b] > best:
                best, idx = float(row[b]), b
        peak_bin[f], peak_db[f] = idx, best
        for b in range(0, min(low_max_bin, nb - 1) + 1):
            if float(row[b]) > low_threshold_db:
                low_bin[f] = b
                break
        for b in range(nb - 1, high_min_bin - 1, -1):
            if float(row[b]) > high_threshold_db:
                high_bin[f] = b
                break
    return peak_bin, peak_db, low_bin, high_bin
