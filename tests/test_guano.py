"""spectrogram_amd.guano against what the reference's own guanoReader.js returned for the same files
(oracle/make_golden_guano.py).  No GPU needed."""
import base64
import json
import os

import pytest

from spectrogram_amd import guano

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_guano.json")))


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_guano_reader_matches_the_reference(c):
    if "input_text" in c:
        assert guano.parse_guano_metadata(c["input_text"]) == c["parsed_text"]
        return
    data = base64.b64decode(c["file"])
    if c["threw"]:
        assert c["threw"] == "RangeError"
        with pytest.raises(guano.GuanoRangeError):
            guano.extract_guano_metadata(data)
        return
    text = guano.extract_guano_metadata(data)
    assert text == c["text"]
    assert guano.parse_guano_metadata(text) == c["parsed"]


def test_no_file():
    assert guano.extract_guano_metadata(None) == "(no file selected)"
    assert guano.parse_guano_metadata(None) == {}


def test_wav_parse_points_at_the_same_chunk():
    # sono_wav_parse reports where the 'guan' chunk's body lies; the reader must find its text there
    import numpy as np
    import spectrogram_amd as sa
    c = next(c for c in CASES if c["name"] == "guano_after_data")
    data = base64.b64decode(c["file"])
    info = sa.wav_info(data)
    body = data[info["guano_offset"]:info["guano_offset"] + info["guano_bytes"]].decode()
    assert body == guano.extract_guano_metadata(data)
