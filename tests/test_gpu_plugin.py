"""spectrogram_amd.plugin.Spectrogram.getFrequencies against golden vectors
captured from the reference plugin's own getFrequencies (oracle/make_golden_plugin.py)."""
import base64
import json
import os

import numpy as np
import pytest

from spectrogram_amd.plugin import AudioBuffer, Spectrogram
from tests import parity

pytestmark = pytest.mark.gpu

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_plugin.json")))


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_get_frequencies_matches_reference(c):
    chans = [np.frombuffer(base64.b64decode(x), np.float32) for x in c["x"]]
    sp = Spectrogram.create({"fftSamples": c["fft"], "windowFunc": c["window"], "noverlap": c["noverlap"],
                             "gainDB": c["gain_db"], "rangeDB": c["range_db"], "splitChannels": c["split_channels"],
                             "peakMode": c["peak_mode"], "peakThreshold": c["peak_threshold"],
                             "canvasWidth": c["canvas_width"], "scale": c["scale"], "frequencyMin": c["frequency_min"],
                             "frequencyMax": c["frequency_max"]})
    frames = sp.getFrequencies(AudioBuffer(chans, c["sr"]))
    assert len(frames) == len(c["frames"])
    bins = c["fft"] // 2
    if c["bank"]:
        # the two-tap filter bank the plugin builds: same taps, weights to float rounding of the scale functions
        want = np.zeros(c["bank"]["size"], np.float32)
        want[np.frombuffer(base64.b64decode(c["bank"]["idx"]), np.uint32)] = np.frombuffer(base64.b64decode(c["bank"]["val"]), np.float32)
        got = sp._filterBankFlat
        assert got.shape == want.shape and np.array_equal(got != 0, want != 0)
        assert np.max(np.abs(got - want)) <= 1e-5 and np.mean(got == want) > 0.999
    else:
        assert sp._filterBankFlat is None
    for got, want in zip(frames, c["frames"]):
        ref = np.frombuffer(base64.b64decode(want["u8"]), np.uint8).reshape(want["count"], bins)
        assert got.shape == ref.shape and got.dtype == np.uint8              # identical frame arrays: count and length
        nd, nt = parity.u8_diff(got, ref, c["name"])
        assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    if c["peak_mode"]:
        assert len(sp.peakBandArrayPerChannel) == len(c["peaks"])
        for ch, (got_ch, ref_ch) in enumerate(zip(sp.peakBandArrayPerChannel, c["peaks"])):
            assert len(got_ch) == len(ref_ch)
            ref_u8 = np.frombuffer(base64.b64decode(c["frames"][ch]["u8"]), np.uint8).reshape(-1, bins)
            same = 0
            for f, (g, r) in enumerate(zip(got_ch, ref_ch)):
                if np.array_equal(frames[ch][f], ref_u8[f]):             # where the u8 row is identical the list must be
                    assert [[p["bin"], p["magnitude"], int(p["isMainPeak"])] for p in g] == r, (c["name"], ch, f)
                    same += 1
            assert same >= 0.66 * len(ref_ch)            # rows without a single 1-LSB difference (about 2e-4 of the codes differ)
    else:
        assert sp.peakBandArrayPerChannel == []
    sp.destroy()
