"""The numeric contract of the reference's wavesurfer spectrogram plugin
(modules/spectrogram.esm.js) over the GPU engine: what `getFrequencies` hands to
the renderer, batCallDetector and peakControl -- per channel, one Uint8Array of
dB codes per frame, plus the peak-mode lists.  Canvas drawing, colour maps and
the DOM stay with the application.
"""
import math

import numpy as np

from . import SpectrogramEngine

__all__ = ["Spectrogram", "AudioBuffer", "SCALES"]

_ERB_A = 1e3 * math.log(10) / 107.939                                # spectrogram.esm.js:344


def _hz_to_bark(t):                                                  # :1251-1256
    e = 26.81 * t / (1960 + t) - .53
    if e < 2:
        e += .15 * (2 - e)
    if e > 20.1:
        e += .22 * (e - 20.1)
    return e


def _bark_to_hz(t):                                                  # :1257-1261
    if t < 2:
        t = (t - .3) / .85
    if t > 20.1:
        t = (t + 4.422) / 1.22
    return (t + .53) / (26.28 - t) * 1960


# scale name -> (Hz -> scale, scale -> Hz), spectrogram.esm.js:1233-1273
SCALES = {
    "mel": (lambda t: 2595 * math.log10(1 + t / 700), lambda t: 700 * (math.pow(10, t / 2595) - 1)),
    "logarithmic": (lambda t: math.log10(max(1, t)), lambda t: math.pow(10, t)),
    "bark": (_hz_to_bark, _bark_to_hz),
    "erb": (lambda t: _ERB_A * math.log10(1 + .00437 * t), lambda t: (math.pow(10, t / _ERB_A) - 1) / .00437),
}


class AudioBuffer:
    """The three members of a WebAudio AudioBuffer that getFrequencies reads."""

    def __init__(self, channels, sample_rate):
        self._channels = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
        self.sampleRate = sample_rate
        self.numberOfChannels = len(self._channels)
        self.length = len(self._channels[0]) if self._channels else 0

    def getChannelData(self, k):
        return self._channels[k]


class Spectrogram:
    """Spectrogram.create(options) of spectrogram.esm.js:405-444, numeric options only:
    fftSamples (512), windowFunc ('hann'), alpha, noverlap (0 = derive it from the canvas width),
    gainDB (20), rangeDB (80), splitChannels (False), peakMode (False), peakThreshold (0.4),
    scale ('mel' as in the plugin; the application passes 'linear', wsManager.js:53), frequencyMin / frequencyMax (0),
    canvasWidth (stands in for this.canvas.width), device."""

    def __init__(self, options=None):
        o = dict(options or {})
        self.options = o
        self.fftSamples = int(o.get("fftSamples", 512))
        self.windowFunc = o.get("windowFunc", "hann")
        self.alpha = o.get("alpha")
        self.noverlap = int(o.get("noverlap", 0) or 0)
        self.gainDB = float(o.get("gainDB", 20))               # spectrogram.esm.js:412
        self.rangeDB = float(o.get("rangeDB", 80))             # :413
        self.canvasWidth = int(o.get("canvasWidth", 800))
        self.frequencyMin = o.get("frequencyMin") or 0         # :410-411
        self.frequencyMax = o.get("frequencyMax") or 0
        self.scale = o.get("scale") or "mel"                   # :414
        self.numMelFilters = self.numLogFilters = self.numBarkFilters = self.numErbFilters = self.fftSamples // 2   # :415-418
        self._filterBankCacheByKey = {}
        self._loadedFilterBankKey = None
        self._filterBankFlat = None
        self.peakBandArrayPerChannel = []
        self._engine = SpectrogramEngine(self.fftSamples, self.windowFunc, self.alpha, device=o.get("device", -1))

    @classmethod
    def create(cls, options=None):
        return cls(options)

    def destroy(self):
        if self._engine is not None:
            self._engine.free()                                # :584-595
            self._engine = None

    def createFilterBank(self, num_filters, sample_rate, hz_to_scale, scale_to_hz):
        """spectrogram.esm.js:1197-1232: num_filters rows of fftSamples/2 + 1 weights, two taps per row (linear
        interpolation between the FFT bins either side of the filter's centre) -> float32 [num_filters, fftSamples/2 + 1]."""
        lo, hi = hz_to_scale(0), hz_to_scale(sample_rate / 2)
        f_min = hz_to_scale(self.frequencyMin) if self.frequencyMin > 0 else lo                               # :1210-1211
        f_max = hz_to_scale(self.frequencyMax) if 0 < self.frequencyMax < sample_rate / 2 else hi
        width = self.fftSamples // 2 + 1
        bank = np.zeros((num_filters, width), np.float32)
        h = sample_rate / self.fftSamples
        for e in range(num_filters):                                                                          # :1222-1229
            s = scale_to_hz(f_min + e / num_filters * (f_max - f_min))
            o = math.floor(s / h)
            left = o * h
            c = (s - left) / ((o + 1) * h - left)
            if 0 <= o < width:
                bank[e, o] = 1 - c
            if 0 <= o + 1 < width:
                bank[e, o + 1] = c
        return bank

    def _select_filter_bank(self, sample_rate):
        """The filter-bank part of getFrequencies (spectrogram.esm.js:1399-1427)."""
        if self.scale != "linear":
            key = f"{self.scale}:{sample_rate}:{self.frequencyMin}:{self.frequencyMax}"
            if self._loadedFilterBankKey != key:
                if key not in self._filterBankCacheByKey:
                    if self.scale not in SCALES:                   # the reference's switch leaves the bank undefined:
                        self._filterBankCacheByKey[key] = None     # flattenAndLoadFilterBank(undefined) clears it
                    else:
                        self._filterBankCacheByKey[key] = self.createFilterBank(self.fftSamples // 2, sample_rate, *SCALES[self.scale])
                bank = self._filterBankCacheByKey[key]
                if bank is None or len(bank) == 0:                                                            # :1319-1327
                    if self._filterBankFlat is not None:
                        self._engine.clear_filter_bank()
                    self._filterBankFlat = None
                else:                                                                                         # :1329-1348
                    self._filterBankFlat = np.ascontiguousarray(bank, np.float32).ravel()
                    self._engine.load_filter_bank(self._filterBankFlat, bank.shape[0])
                self._loadedFilterBankKey = key
        elif self._loadedFilterBankKey is not None:
            if self._filterBankFlat is not None:
                self._engine.clear_filter_bank()
            self._filterBankFlat = None
            self._loadedFilterBankKey = None

    def getFrequencies(self, buffer):
        """spectrogram.esm.js:1364-1529 -> [channel][frame] -> np.uint8[fftSamples/2] (a frames x bins array per channel;
        bins = the number of filters on a non-linear scale); fills peakBandArrayPerChannel when options.peakMode."""
        if self._engine is None or buffer is None:
            return None
        r = self.fftSamples
        self.frequencyMax = self.frequencyMax or buffer.sampleRate / 2                                        # :1377
        self._select_filter_bank(buffer.sampleRate)
        channels = buffer.numberOfChannels if self.options.get("splitChannels") else 1          # :1374
        o = self.noverlap
        if not o:                                                                             # :1384-1388
            e = buffer.length / self.canvasWidth
            o = max(math.floor(r * 0.05), math.floor(r - e + 0.5))
        self.peakBandArrayPerChannel = []
        slider = self.options.get("peakThreshold", 0.4)                                       # :1432-1433
        effective_threshold = 0.60 + (math.pow(slider, 1.5) * 0.39)
        peak_mode = bool(self.options.get("peakMode"))
        out = []
        for ch in range(channels):
            u8 = self._engine.compute_spectrogram_u8(buffer.getChannelData(ch), o, self.gainDB, self.rangeDB)   # :1450
            global_max = self._engine.get_global_max()                                        # :1476
            noise_floor = global_max * 0.063                                                  # :1477
            frame_max = self._engine.get_peak_magnitudes(0.0)                                 # :1479
            filters = self._engine.get_num_filters()                                          # :1481-1483
            bins = filters if self.scale != "linear" and filters > 0 else r // 2
            frames = u8[: (u8.size // bins) * bins].reshape(-1, bins)                          # :1483-1488
            out.append(frames)
            if peak_mode:
                lists = []
                local_max = frames.max(axis=1) if frames.size else np.zeros(0, np.uint8)
                for f in range(frames.shape[0]):
                    if frame_max[f] < noise_floor:                                            # :1493-1496 (f32 vs f64 product)
                        lists.append([])
                        continue
                    lm = int(local_max[f])
                    peaks = []
                    if lm > 10:                                                               # :1506
                        row = frames[f]
                        for k in np.nonzero(row >= lm * effective_threshold)[0]:              # :1503-1515
                            peaks.append({"bin": int(k), "magnitude": int(row[k]), "isMainPeak": bool(row[k] == lm)})
                    lists.append(peaks)
                self.peakBandArrayPerChannel.append(lists)
        return out
