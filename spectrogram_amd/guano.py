"""The reference's GUANO metadata reader (modules/guanoReader.js) for batch use: the text of the first RIFF chunk that
holds a GUANO header, and the four fields the application takes from it.  Host only.
"""
import re
import struct

import numpy as np

__all__ = ["extract_guano_metadata", "parse_guano_metadata", "GuanoRangeError"]

NO_FILE = "(no file selected)"
NO_GUANO = "(No GUANO metadata found in file)"
_JS_SPACE = "\t\n\x0b\x0c\r \xa0                　﻿"
_FLOAT_PREFIX = re.compile(r"[+-]?(?:Infinity|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")


class GuanoRangeError(ValueError):
    """Where the reference's DataView / Uint8Array raise a RangeError: a chunk header or body that runs past the end
    of the file (guanoReader.js:13-21)."""


def extract_guano_metadata(file_bytes):
    """extractGuanoMetadata (guanoReader.js:3-34).  file_bytes: the whole file (None = no file).  Walks the chunks from
    byte 12 on, decodes EVERY chunk's body as UTF-8 and returns the first that contains 'GUANO|Version:'."""
    if file_bytes is None:
        return NO_FILE
    buf = bytes(file_bytes)
    n, pos = len(buf), 12
    while pos < n:
        if pos + 8 > n:
            raise GuanoRangeError("chunk header past the end of the file")
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        if pos + 8 + size > n:
            raise GuanoRangeError("chunk body past the end of the file")
        text = buf[pos + 8:pos + 8 + size].decode("utf-8", errors="replace")
        if text.startswith("﻿"):                    # TextDecoder drops a leading byte-order mark
            text = text[1:]
        if "GUANO|Version:" in text:
            return text
        pos += 8 + size + (size & 1)                     # word alignment (:29-30)
    return NO_GUANO


def _number_to_string(x):
    """Number.prototype.toString for a finite double."""
    if x == 0:
        return "0"
    a = abs(x)
    if 1e-7 <= a < 1e21:
        return np.format_float_positional(x, unique=True, trim="-")
    m, e = repr(float(x)).split("e")                     # shortest round-trip digits, as in JavaScript
    if m.endswith(".0"):
        m = m[:-2]
    return f"{m}e{'+' if int(e) >= 0 else '-'}{abs(int(e))}"


def _parse_float(s):
    """parseFloat: the longest numeric prefix after leading white space, else NaN."""
    m = _FLOAT_PREFIX.match(s.lstrip(_JS_SPACE))
    if not m:
        return float("nan")
    t = m.group(0)
    return float(t.replace("Infinity", "inf"))


def parse_guano_metadata(text):
    """parseGuanoMetadata (guanoReader.js:36-80) -> {} or {'date', 'time', 'latitude', 'longitude'} (strings)."""
    if not text or text.startswith("(No GUANO"):
        return {}
    meta = {}
    for line in re.split(r"\r?\n", text):
        i = line.find(":")
        if i == -1:
            continue
        meta[line[:i].strip(_JS_SPACE)] = line[i + 1:].strip(_JS_SPACE)
    date = time = lat = lon = ""
    ts = meta.get("Timestamp")
    if ts:                                               # :48-54
        parts = ts.split(" ")
        date_part = parts[0]
        time_part = (parts[1] if len(parts) > 1 else "").split("+")[0]
        date = date_part.replace("-", "/")
        time = time_part[:5].replace(":", "", 1)
    pos = meta.get("Loc Position")
    if pos:                                              # :56-72
        fields = re.split("[" + _JS_SPACE + "]+", pos)
        lat = fields[0]
        if len(fields) > 1 and fields[1]:
            v = _parse_float(fields[1])
            if v != v:
                lon = fields[1]
            else:
                if v < 0 and 113 <= abs(v) <= 115:       # a sign error some recorders make around 114 E (:62-64)
                    v = abs(v)
                lon = "Infinity" if v == float("inf") else "-Infinity" if v == float("-inf") else _number_to_string(v)
    return {"date": date, "time": time, "latitude": lat, "longitude": lon}
