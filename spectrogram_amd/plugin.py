"""The numeric contract of the reference's wavesurfer spectrogram plugin
(modules/spectrogram.esm.js) over the GPU engine: what `getFrequencies` hands to
the renderer, batCallDetector and peakControl -- per channel, one Uint8Array of
dB codes per frame, plus the peak-mode lists.  Canvas drawing, colour maps and
the DOM stay with the application.
"""
import math

import numpy as np

from . import SpectrogramEngine

__all__ = ["Spectrogram", "AudioBuffer"]


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
        self.peakBandArrayPerChannel = []
        self._engine = SpectrogramEngine(self.fftSamples, self.windowFunc, self.alpha, device=o.get("device", -1))

    @classmethod
    def create(cls, options=None):
        return cls(options)

    def destroy(self):
        if self._engine is not None:
            self._engine.free()                                # :584-595
            self._engine = None

    def getFrequencies(self, buffer):
        """spectrogram.esm.js:1364-1529 for scale 'linear' -> [channel][frame] -> np.uint8[fftSamples/2]
        (a frames x bins array per channel); fills peakBandArrayPerChannel when options.peakMode."""
        if self._engine is None or buffer is None:
            return None
        r = self.fftSamples
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
        bins = r // 2
        for ch in range(channels):
            u8 = self._engine.compute_spectrogram_u8(buffer.getChannelData(ch), o, self.gainDB, self.rangeDB)   # :1450
            global_max = self._engine.get_global_max()                                        # :1476
            noise_floor = global_max * 0.063                                                  # :1477
            frame_max = self._engine.get_peak_magnitudes(0.0)                                 # :1479
            frames = u8.reshape(-1, bins)                                                     # :1483-1488
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
