"""Host parts of the plugin mirror that need no GPU: the two-tap filter banks of the non-linear frequency scales
(createFilterBank and the scale functions of spectrogram.esm.js:1197-1273) against the banks the reference's own code
built during the golden capture (oracle/make_golden_plugin.py)."""
import base64
import json
import os

import numpy as np

from spectrogram_amd.plugin import SCALES, Spectrogram

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_plugin.json")))


def test_filter_banks_match_the_reference_bit_for_bit():
    seen = set()
    for c in CASES:
        if not c["bank"]:
            continue
        sp = Spectrogram.__new__(Spectrogram)                       # no engine: only the fields the builder reads
        sp.fftSamples = c["fft"]
        sp.frequencyMin = c["frequency_min"] or 0
        sp.frequencyMax = c["frequency_max"] or c["sr"] / 2          # getFrequencies sets it before building (:1377)
        got = sp.createFilterBank(c["fft"] // 2, c["sr"], *SCALES[c["scale"]])
        want = np.zeros(c["bank"]["size"], np.float32)
        want[np.frombuffer(base64.b64decode(c["bank"]["idx"]), np.uint32)] = np.frombuffer(base64.b64decode(c["bank"]["val"]), np.float32)
        assert got.shape == (c["fft"] // 2, c["fft"] // 2 + 1)
        assert np.array_equal(got.ravel().view(np.uint32), want.view(np.uint32)), c["name"]
        assert ((got != 0).sum(axis=1) <= 2).all()
        seen.add(c["scale"])
    assert seen == {"mel", "logarithmic", "bark", "erb"}


def test_scale_functions_invert_each_other():
    for name, (to_scale, to_hz) in SCALES.items():
        for hz in (50.0, 1000.0, 20000.0, 96000.0):
            assert abs(to_hz(to_scale(hz)) - hz) <= 1e-6 * hz, (name, hz)
