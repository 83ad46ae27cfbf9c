"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python restatement of the call-framing stage of the reference's detector
(/root/reference/modules/batCallDetector.js): what detectCalls does between
detectCallSegments and measureFrequencyParameters, and the helpers it uses.
Only tests/ may import it.  Scalar arithmetic is done in Python floats (IEEE
doubles, like JavaScript numbers); values that the reference keeps in a
Float32Array are rounded through numpy.float32 at the same places.

Parity status: PINNED by tests/golden/golden_framing.npz, captured by
oracle/make_golden_framing.py from the reference's own JavaScript methods.
"""
import math

import numpy as np

from oracle import detector_oracle as do


def savitzky_golay(data):
    """savitzkyGolay, batCallDetector.js:1610-1640 (window 5, order 2)."""
    data = [float(v) for v in data]
    if len(data) < 5:
        return data                                            # :1611
    coeffs = [-3, 12, 17, 12, -3]                              # :1618
    out = []
    for i in range(len(data)):
        s = 0.0
        for j in range(-2, 3):
            if 0 <= i + j < len(data):
                s += data[i + j] * coeffs[j + 2]               # :1631
        out.append(s / 35)                                     # :1636: always / 35, also at the edges
    return out


def refine_end_using_oscillogram(audio, sample_rate, start_sample, end_sample):
    """refineEndUsingOscillogram, batCallDetector.js:1646-1803 -> refined end sample."""
    audio = np.asarray(audio, np.float32)
    safe_start = max(0, start_sample)                          # :1650
    safe_end = min(len(audio), end_sample)
    if safe_end - safe_start < sample_rate * 0.0005:           # :1653
        return end_sample
    window = math.floor(sample_rate * (0.1 / 1000))            # :1660
    sustained = math.floor(sample_rate * (0.5 / 1000))         # :1663
    hop = math.floor(window / 2)                               # :1664
    db_values, centres = [], []
    peak, peak_index = -math.inf, 0
    i = safe_start
    while i < safe_end - window:                               # :1676
        seg = audio[i:i + window].astype(np.float64)
        sum_sq = 0.0
        for v in seg:                                          # sequential f64 sum (:1678-1681)
            sum_sq += v * v
        db = 20 * math.log10(math.sqrt(sum_sq / window) + 1e-9)
        if db > peak:                                          # on the unrounded value (:1688)
            peak, peak_index = db, len(db_values)
        db_values.append(float(np.float32(db)))                # Float32Array (:1669,1685)
        centres.append(i + math.floor(window / 2))
        i += hop
    steps = len(db_values)
    if steps == 0:
        return end_sample
    min_db, min_index = peak, peak_index                       # :1699-1700
    for i in range(peak_index + 1, steps):
        cur = db_values[i]
        if cur < min_db:
            min_db, min_index = cur, i
        if min_db < -60 and cur < -60 + 2:                     # :1714
            return centres[min_index]
        if cur - min_db > 0.5:                                 # :1722-1723
            if min_db > -32:                                   # :1728: still inside the signal body
                continue
            limit = min(steps, i + math.ceil(sustained / hop)) # :1744
            if all(db_values[k] >= min_db + 0.5 for k in range(i + 1, limit)):
                return centres[min_index]                      # :1798
    return end_sample


def zonal_noise_floors(power_matrix, freq_bins, start_frame, end_frame):
    """calculateZonalNoiseFloors, batCallDetector.js:1818-1924 -> list of floors by 10 kHz zone index."""
    p = np.asarray(power_matrix, np.float32)
    fb = np.asarray(freq_bins, np.float32).astype(np.float64)
    num_zones = math.floor(fb[-1] / 10000) + 1                 # :1837-1839
    hist = np.zeros((num_zones, 141), np.int64)                # -120 .. +20 dB (:1829-1831)
    zone = np.floor(fb / 10000).astype(np.int64)
    for f in range(start_frame, end_frame + 1):
        if f >= p.shape[0]:
            break
        row = np.maximum(p[f].astype(np.float64), -100.0)      # :1857
        idb = np.clip(np.floor(row), -120, 20).astype(np.int64) + 120
        for b in range(len(row)):
            if zone[b] < num_zones:
                hist[zone[b], idb[b]] += 1
    out = []
    for z in range(num_zones):
        if hist[z].any():
            out.append(float(int(np.argmax(hist[z])) - 120) - 2.0)    # first maximum = lowest dB on ties (:1907)
        else:
            out.append(-102.0)                                 # :1919
    return out


def noise_floor_percentile(power_matrix):
    """The 25th-percentile value detectCalls takes as noise floor (batCallDetector.js:1223-1233), before the -80 floor."""
    v = np.sort(np.asarray(power_matrix, np.float32).ravel())
    return float(v[max(0, math.floor(len(v) * 0.25))])


def scan_spectrum_for_frequency(row, freq_bins, threshold_db, high_to_low):
    """_scanSpectrumForFrequency, batCallDetector.js:2936-2987 -> (freq_Hz, bin) or (-1, -1)."""
    row = [float(v) for v in np.asarray(row, np.float32)]
    fb = [float(v) for v in np.asarray(freq_bins, np.float32)]
    n = len(row)
    order = range(n - 1, -1, -1) if high_to_low else range(n)
    for b in order:
        if row[b] > threshold_db:
            freq = fb[b]
            nb = b + 1 if high_to_low else b - 1
            if 0 <= nb < n and row[nb] < threshold_db:
                ratio = (row[b] - threshold_db) / (row[b] - row[nb])
                freq = fb[b] + ratio * (fb[nb] - fb[b]) if high_to_low else fb[b] - ratio * (fb[b] - fb[nb])
            return freq, b
    return -1.0, -1


def auto_highpass_khz(peak_khz):
    """calculateAutoHighpassFilterFreq, batCallDetector.js:4028-4035."""
    return 30 if peak_khz >= 40 else 25 if peak_khz >= 35 else 20 if peak_khz >= 30 else 0


def highpass_filter(audio, filter_hz, sample_rate, order=4):
    """applyHighpassFilter + _applyButterworthStage, batCallDetector.js:4045-4140 -> float32 array."""
    audio = np.asarray(audio, np.float32)
    if len(audio) == 0 or filter_hz <= 0:
        return audio
    order = max(1, min(8, math.floor(order + 0.5)))            # Math.round (:4051)
    norm = filter_hz / (sample_rate / 2)
    if norm >= 1:                                              # :4058
        return audio
    wc = math.tan(math.pi * norm / 2)                          # :4063
    out = audio.copy()
    for _ in range(order // 2):
        out = _stage(out, wc, 2)
    if order % 2:
        out = _stage(out, wc, 1)
    return out


def _stage(x, wc, order):
    res = np.empty(len(x), np.float32)
    wc2 = wc * wc
    if order == 1:
        den = wc + 1
        b0, b1, a1 = 1 / den, -1 / den, (wc - 1) / den
        x1 = y1 = 0.0
        for i, v in enumerate(x.astype(np.float64)):
            y0 = b0 * v + b1 * x1 - a1 * y1                    # :4109
            res[i] = y0
            x1, y1 = v, y0                                     # the recurrence keeps the unrounded double
        return res
    s2 = math.sqrt(2) * wc
    den = wc2 + s2 + 1
    b0, b1, b2 = 1 / den, -2 / den, 1 / den
    a1, a2 = (2 * (wc2 - 1)) / den, (wc2 - s2 + 1) / den
    x1 = x2 = y1 = y2 = 0.0
    for i, v in enumerate(x.astype(np.float64)):
        y0 = b0 * v + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2    # :4131
        res[i] = y0
        x2, x1, y2, y1 = x1, v, y1, y0
    return res


def frame_calls(audio, sample_rate, fft_size, power_matrix, time_frames, segments, min_call_duration_ms=1.0,
                padding_ms=3, full_file_rules=False):
    """The per-segment part of detectCalls, batCallDetector.js:1128-1208 (or of processFullFile, :739-810, with
    full_file_rules and padding_ms=5) -> list of dicts {startFrame, endFrame, startTime_s, endTime_s, duration_ms}."""
    p = np.asarray(power_matrix, np.float32)
    tf = [float(t) for t in time_frames]
    frames = p.shape[0]
    if frames < 2:
        return []                                              # the reference has no defined behaviour here
    frame_sec = 1 / (sample_rate / fft_size)                   # :1130
    out = []
    for s, e in segments:
        if not full_file_rules and (e - s + 1) * frame_sec * 1000 < min_call_duration_ms:      # :1131-1133
            continue
        per_frame = tf[1] - tf[0]
        pad = math.ceil((padding_ms / 1000) / per_frame)       # :1146-1148, :743-745
        safe_start = max(0, s - pad)
        safe_end = min(frames - 1, e + pad)
        start_sample = math.floor(tf[safe_start] * sample_rate)        # :1158-1159
        end_sample = math.floor(tf[safe_end] * sample_rate)
        refined = refine_end_using_oscillogram(audio, sample_rate, start_sample, end_sample)
        if refined < end_sample:
            t_end = refined / sample_rate
            new_end = safe_end
            while new_end > safe_start and tf[new_end] > t_end:        # :1172
                new_end -= 1
            safe_end = min(frames - 1, new_end + 1)            # :1188
        t0 = tf[safe_start]
        t1 = tf[min(safe_end + 1, len(tf) - 1)]                # :1197
        dur = (t1 - t0) * 1000                                 # BatCall.calculateDuration, :161
        if dur <= 1.0 or (not full_file_rules and dur < min_call_duration_ms):     # :1206, :807
            continue
        out.append(dict(startFrame=safe_start, endFrame=safe_end, startTime_s=t0, endTime_s=t1, duration_ms=dur))
    return out


def detect_calls_framing(engine, audio, sample_rate, flow_khz, fhigh_khz, hop_percent=3.125, call_threshold_db=-24.0,
                         min_call_duration_ms=1.0, noise_audio=None):
    """detectCalls (batCallDetector.js:1070-1289) without the measurement stage:
    -> (calls, zonal noise floors, noise floor dB, spectrogram dict)."""
    sp = do.generate_spectrogram(engine, audio, sample_rate, flow_khz, fhigh_khz, hop_percent)
    if sp is None:
        return [], [], None, None
    p, tf, fb = sp["powerMatrix"], sp["timeFrames"], sp["freqBins"]
    ns = do.generate_spectrogram(engine, noise_audio, sample_rate, flow_khz, fhigh_khz, hop_percent) \
        if noise_audio is not None else None
    if ns is not None and len(ns["powerMatrix"]) > 0:          # :1086
        total = len(ns["powerMatrix"])
        first = 0
        if len(ns["timeFrames"]) > 1:
            per = float(ns["timeFrames"][1] - ns["timeFrames"][0])
            first = max(0, total - math.ceil(0.005 / per))     # :1093-1098
        zonal = zonal_noise_floors(ns["powerMatrix"], ns["freqBins"], first, total - 1)
    else:
        zonal = zonal_noise_floors(p, fb, 0, min(5, len(p) - 1))       # :1115-1120
    segs = do.detect_call_segments(p, call_threshold_db)
    calls = frame_calls(audio, sample_rate, engine.get_fft_size(), p, tf, segs, min_call_duration_ms) if segs else []
    floor = max(noise_floor_percentile(p), -80.0)              # :1232-1235
    return calls, zonal, floor, sp


def process_full_file(engine, audio, sample_rate, flow_khz, fhigh_khz, threshold_db=-60, padding_ms=5, hop_percent=3.125,
                      call_threshold_db=-24.0, min_call_duration_ms=1.0, keep_slices=False):
    """processFullFile, batCallDetector.js:584-840, without the measurement stage ->
    list of dicts {startTime_s, endTime_s, duration_ms, peakPower_dB, frames, highpass_kHz, roi} in the reference's order
    (per region: strongest call first).  keep_slices: every framed call, clicks included, also carries what the
    reference hands to measureFrequencyParameters (:786-797): spectrogram, timeFrames, freqBins, freqResolution,
    zonalNoiseMap, its region-relative times (relStart_s, relEnd_s) and the region's offset_s."""
    audio = np.ascontiguousarray(audio, np.float32)
    raw = do.fast_scan_segments(engine, audio, sample_rate, flow_khz, fhigh_khz, threshold_db)        # :594
    if not raw:
        return []
    merged = do.merge_and_pad_segments(raw, len(audio), sample_rate, padding_ms)                      # :601
    out = []
    for roi, (s0, s1) in enumerate(merged):
        seg_audio = audio[s0:s1]
        sp = do.generate_spectrogram(engine, seg_audio, sample_rate, flow_khz, fhigh_khz, hop_percent)
        if sp is None:
            continue
        p, tf, fb = sp["powerMatrix"], sp["timeFrames"], sp["freqBins"]
        floors = zonal_noise_floors(p, fb, 0, min(5, len(p) - 1))                                    # :622-627, before the high-pass
        flat = int(np.argmax(p))                                                                     # first maximum, :635-644
        peak_khz = float(fb[flat % p.shape[1]]) / 1000                                               # :646
        cutoff = auto_highpass_khz(peak_khz)
        if cutoff > 0:                                                                               # :651-671
            seg_audio = highpass_filter(seg_audio, cutoff * 1000, sample_rate)
            sp2 = do.generate_spectrogram(engine, seg_audio, sample_rate, flow_khz, fhigh_khz, hop_percent)
            if sp2 is not None:
                p, tf = sp2["powerMatrix"], sp2["timeFrames"]
        segs = do.detect_call_segments(p, call_threshold_db)                                         # :677
        cands = []
        for a, b in segs:
            if tf[b] - tf[a] < min_call_duration_ms / 1000:                                          # :688-689
                continue
            block = p[a:b + 1]
            k = int(np.argmax(block))                                                                # first maximum, :695-703
            cands.append(dict(seg=(a, b), peakPower=float(block.ravel()[k]), peakTime=float(tf[a + k // p.shape[1]])))
        cands.sort(key=lambda c: -c["peakPower"])                                                    # stable, :714
        kept = []
        for c in cands:
            if all(abs(c["peakTime"] - k["peakTime"]) >= 0.030 for k in kept):                       # :720-734
                kept.append(c)
        offset = s0 / sample_rate                                                                    # :737
        for c in kept:
            fr = frame_calls(seg_audio, sample_rate, engine.get_fft_size(), p, tf, [c["seg"]], min_call_duration_ms,
                             padding_ms=5, full_file_rules=True)                                     # :743-810
            for k in fr:
                blk = p[k["startFrame"]:k["endFrame"] + 1]
                row = blk[int(np.argmax(blk)) // p.shape[1]].astype(np.float64)                      # the peak frame
                above = np.flatnonzero(row > float(blk.max()) - 12)                                  # :3061-3073
                click = len(above) and (above[-1] - above[0]) * sp["freqResolution"] / 1000 > 20.0   # :3075-3087
                if click and not keep_slices:
                    continue
                out.append(dict(startTime_s=k["startTime_s"] + offset, endTime_s=k["endTime_s"] + offset,
                                duration_ms=k["duration_ms"], peakPower_dB=float(blk.max()),
                                frames=k["endFrame"] - k["startFrame"] + 1, highpass_kHz=cutoff, roi=roi))
                if keep_slices:
                    out[-1].update(spectrogram=blk, timeFrames=tf[k["startFrame"]:k["endFrame"] + 2], freqBins=fb,
                                   freqResolution=sp["freqResolution"], relStart_s=k["startTime_s"], relEnd_s=k["endTime_s"],
                                   zonalNoiseMap={10 * z: v for z, v in enumerate(floors)}, offset_s=offset)
    return out


def rms_snr(power_matrix, freq_bins, start_idx, end_idx, low_khz, high_khz, flow_khz, fhigh_khz, noise_matrix=None,
            noise_freq_bins=None):
    """calculateRMSbasedSNR, batCallDetector.js:402-571 ->
    dict(snr_dB, signalPowerMean_dB, noisePowerMean_dB, signalCount, noiseCount, threshold, max, min); None where the
    reference leaves null."""
    p = np.asarray(power_matrix, np.float32).astype(np.float64)
    fb = np.asarray(freq_bins, np.float32).astype(np.float64)
    out = dict(snr_dB=None, signalPowerMean_dB=None, noisePowerMean_dB=None, signalCount=0, noiseCount=0, threshold=None,
               max=None, min=None)
    lo, hi = low_khz * 1000, high_khz * 1000                               # :423-424
    in_f = (fb >= lo) & (fb <= hi)
    t1 = min(end_idx, p.shape[0] - 1)
    in_t = np.zeros(p.shape[0], bool)
    if t1 >= start_idx:
        in_t[max(start_idx, 0):t1 + 1] = True
    box = in_t[:, None] & in_f[None, :]
    if not box.any():                                                      # :452
        return out
    smax, smin = float(p[box].max()), float(p[box].min())
    thr = smin + (smax - smin) * 0.25                                      # :460-462
    out.update(threshold=thr, max=smax, min=smin)
    sig = box & (p > thr)
    signal_sum = float(np.sum(10.0 ** (p[sig] / 10)))                      # :479
    signal_count = int(sig.sum())
    if noise_matrix is not None and len(noise_matrix) > 0:                 # :496-516
        q = np.asarray(noise_matrix, np.float32).astype(np.float64)
        nfb = np.asarray(noise_freq_bins, np.float32).astype(np.float64)
        sel = (nfb >= flow_khz * 1000) & (nfb <= fhigh_khz * 1000)
        noise_sum = float(np.sum(10.0 ** (q[:, sel] / 10)))
        noise_count = int(sel.sum()) * q.shape[0]
    else:                                                                  # :522-538
        noise_sum = float(np.sum(10.0 ** (p[~box] / 10)))
        noise_count = int((~box).sum())
    if signal_count == 0:                                                  # :543
        return out
    if noise_count == 0:                                                   # :548
        out["snr_dB"] = math.inf
        return out
    s_mean, n_mean = signal_sum / signal_count, noise_sum / noise_count
    out.update(signalPowerMean_dB=10 * math.log10(max(s_mean, 1e-16)), noisePowerMean_dB=10 * math.log10(max(n_mean, 1e-16)),
               signalCount=signal_count, noiseCount=noise_count)
    out["snr_dB"] = math.inf if n_mean < 1e-16 else 10 * math.log10(s_mean / n_mean)       # :562-568
    return out
