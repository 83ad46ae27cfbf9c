#!/usr/bin/env python3
"""Case definitions (small synthetic WAV files and texts) of tests/golden/golden_guano.json:
the GUANO metadata reader (guanoReader.js).  TEST INFRASTRUCTURE ONLY.  The expected
outputs were recorded once from the reference's own two functions; the fixture is frozen
data and is never regenerated -- nothing here loads or executes anything that ships inside
the reference.  build_cases() is the record of the inputs.
"""
import base64
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "golden_guano.json")


def chunk(tag, body, pad=True):
    return tag + struct.pack("<I", len(body)) + body + (b"\0" if pad and len(body) % 2 else b"")


def wav(chunks):
    body = b"WAVE" + b"".join(chunks)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def build_cases():
    fmt = chunk(b"fmt ", struct.pack("<HHIIHH", 1, 1, 384000, 768000, 2, 16))
    data = chunk(b"data", bytes(range(64)))
    odd = chunk(b"junk", b"abc")                                       # odd size: one pad byte follows
    g1 = ("GUANO|Version: 1.0\nTimestamp: 2023-07-14 21:05:33+08:00\nLoc Position: 22.4173 114.1257\nMake: Acme\n"
          "Note: a: b : c\nno colon here\n")
    g2 = "GUANO|Version:1.0\r\nTimestamp: 2024-01-02T03:04:05\r\nLoc Position: 22.5 -114.25\r\nSpecies Manual ID:  Pipabr \r\n"
    g3 = "GUANO|Version: 1.0\nLoc Position: 51.5\nTimestamp:\n"
    g4 = "GUANO|Version: 1.0\nLoc Position: -33.9 -70.5\nTimestamp: 2022-12-31 23:59\n"
    g5 = "GUANO|Version: 1.0\nLoc Position: 22.3 east\nTimestamp: 2021-05-06 07:08:09.5Z\n"
    g6 = "GUANO|Version: 1.0\nLoc Position: 1.5   -113   extra\nTimestamp: 2020-02-29 12:00:00-05:00\nNote: café 蝙蝠\n"
    cases = [
        ("guano_after_data", wav([fmt, data, chunk(b"guan", g1.encode())])),
        ("guano_before_data_crlf", wav([fmt, odd, chunk(b"guan", g2.encode()), data])),
        ("no_guano", wav([fmt, data])),
        ("latitude_only", wav([fmt, chunk(b"guan", g3.encode()), data])),
        ("western_longitude_kept", wav([fmt, data, chunk(b"guan", g4.encode())])),
        ("longitude_not_a_number", wav([fmt, data, chunk(b"guan", g5.encode())])),
        ("longitude_flipped_integer_utf8", wav([fmt, data, chunk(b"guan", g6.encode())])),
        ("text_in_another_chunk", wav([fmt, chunk(b"LIST", b"xx GUANO|Version: 1.0\nTimestamp: 2019-01-01 00:00:00\n"), data])),
        ("invalid_utf8_before", wav([fmt, chunk(b"bext", b"\xff\xfe\xc3("), chunk(b"guan", g3.encode())])),
        ("truncated_chunk", wav([fmt, data])[:-10]),
        ("trailing_bytes", wav([fmt, data]) + b"\x01\x02\x03"),
        ("header_only", wav([])[:12]),
        ("empty_file", b""),
    ]
    out = [dict(name=n, file=base64.b64encode(b).decode()) for n, b in cases]
    texts = [("parse_empty", ""), ("parse_no_guano", "(No GUANO metadata found in file)"), ("parse_no_file", "(no file selected)"),
             ("parse_spaces", "Timestamp:   2023-01-02   03:04:05  \nLoc Position:   10.0    20.0  "),
             ("parse_big_longitude", "Loc Position: 0 123456789012345678901234\nTimestamp: 2023-1-2 3:4"),
             ("parse_exponent", "Loc Position: 5 -1.14e2\nTimestamp: x")]
    for n, t in texts:
        out.append(dict(name=n, file="", input_text=t))
    return out


def main():
    """The fixture is frozen data: nothing here executes anything that ships inside the
    reference.  build_cases() remains as the record of the inputs the fixture was made from."""
    cases = build_cases()
    n = len(cases[0]) if isinstance(cases, tuple) else len(cases)
    print(f"{n} case definitions; tests/golden/golden_guano.json is frozen and is not regenerated "
          "(its expected outputs were recorded once from the reference's own code paths; see the module docstring)")
    return 1


if __name__ == "__main__":
    sys.exit(main())
