"""CPU restatement of the reference's WAV header sniff -- TEST INFRASTRUCTURE ONLY (only tests/ may import it).

Follows modules/fileLoader.js:7-63 statement by statement: `get_wav_sample_rate` is getWavSampleRate (:7-27),
`get_wav_duration_parts` is getWavDuration (:29-63) returning the values it derives the duration from.  Reads
past the end of the buffer raise IndexError where the reference's DataView raises a RangeError.

Parity status: UNPINNED.  The reference holds no fixture for this code and nothing of it was run; the expected
values in tests/test_wav.py follow from reading it.
"""
import struct


def _u8(b, p):
    if p < 0 or p >= len(b):
        raise IndexError("offset outside the buffer")           # DataView: RangeError
    return b[p]


def _u16(b, p):
    if p < 0 or p + 2 > len(b):
        raise IndexError("offset outside the buffer")
    return struct.unpack_from("<H", b, p)[0]


def _u32(b, p):
    if p < 0 or p + 4 > len(b):
        raise IndexError("offset outside the buffer")
    return struct.unpack_from("<I", b, p)[0]


def _chunk_id(b, pos):
    return bytes(_u8(b, pos + i) for i in range(4))             # fileLoader.js:13-18


def get_wav_sample_rate(b):
    """fileLoader.js:7-27: the rate of the FIRST 'fmt ' chunk; 256000 when the walk finds none."""
    pos = 12                                                    # :11
    while pos < len(b) - 8:                                     # :12
        cid = _chunk_id(b, pos)
        size = _u32(b, pos + 4)                                 # :19
        if cid == b"fmt ":
            return _u32(b, pos + 12)                            # :20-22
        pos += 8 + size                                         # :23
        if size % 2 == 1:
            pos += 1                                            # :24
    return 256000                                               # :26


def get_wav_duration_parts(b):
    """fileLoader.js:29-63: {sample_rate, channels, bits_per_sample, data_size, data_pos, duration}.  Every 'fmt '
    chunk in front of the first 'data' chunk overwrites the format; the walk stops at that 'data' chunk, whose
    DECLARED size counts even when the file is shorter."""
    pos, sample_rate, channels, bits, data_size, data_pos = 12, 0, 1, 16, 0, None      # :33-37
    while pos < len(b) - 8:                                     # :38
        cid = _chunk_id(b, pos)
        size = _u32(b, pos + 4)                                 # :45
        if cid == b"fmt ":
            channels = _u16(b, pos + 10)                        # :47
            sample_rate = _u32(b, pos + 12)                     # :48
            bits = _u16(b, pos + 22)                            # :49
        elif cid == b"data":
            data_size, data_pos = size, pos + 8                 # :51
            break                                               # :52
        pos += 8 + size                                         # :54
        if size % 2 == 1:
            pos += 1                                            # :55
    duration = 0.0
    if sample_rate > 0 and data_size > 0:                       # :57
        bytes_per_sample = (bits / 8) * channels                # :58
        duration = (data_size / bytes_per_sample) / sample_rate   # :59-60
    return dict(sample_rate=sample_rate, channels=channels, bits_per_sample=bits, data_size=data_size, data_pos=data_pos,
                duration=duration)
