"""Deterministic synthetic ultrasonic test signal (bat-pass-like FM pulse train).

sample i of stream s:   x = 0.5 * sin(phi(i)) * [inside a pulse] + 0.002 * (u(s, i) - 0.5)
  * pulses: downward linear FM 80 kHz -> 30 kHz, 5 ms long, one every 100 ms;
  * u: counter-based hash of (seed, stream, i) in [0, 1), so any slice of any
    stream can be produced independently, on the host (numpy) or in HBM (torch)
    with identical values -- large inputs never have to cross PCIe.
Used by bench.py and the tests only.
"""
import numpy as np

PULSE_F0 = 80000.0
PULSE_F1 = 30000.0
PULSE_SECONDS = 0.005
PERIOD_SECONDS = 0.1
AMPLITUDE = 0.5
NOISE = 0.002


def _mix32_np(v):
    v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(16)
    v = (v * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(15)
    v = (v * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(16)
    return v


def chirp_numpy(start, count, sample_rate, stream=0, seed=1):
    """Samples [start, start+count) of `stream` as float32 (host)."""
    i = np.arange(start, start + count, dtype=np.int64)
    period = int(round(PERIOD_SECONDS * sample_rate))
    pulse = int(round(PULSE_SECONDS * sample_rate))
    r = i % period
    tt = r.astype(np.float64) / sample_rate
    k = (PULSE_F1 - PULSE_F0) / PULSE_SECONDS
    phase = 2.0 * np.pi * (PULSE_F0 * tt + 0.5 * k * tt * tt)
    sig = np.where(r < pulse, AMPLITUDE * np.sin(phase), 0.0)
    key = (int(seed) * 0x9E3779B1 + int(stream) * 0x85EBCA77) & 0xFFFFFFFF
    h = _mix32_np((i & 0xFFFFFFFF).astype(np.uint64) ^ np.uint64(key))
    h = _mix32_np(h + ((i >> 32).astype(np.uint64) & np.uint64(0xFFFFFFFF)))
    u = (h >> np.uint64(8)).astype(np.float64) / float(1 << 24)
    return (sig + NOISE * (u - 0.5)).astype(np.float32)


def chirp_torch(start, count, sample_rate, stream=0, seed=1, device="cuda", out=None, chunk=1 << 26):
    """Same samples produced on `device` (float32 torch tensor); chunked so that
    multi-gigasample streams need only ~1 GiB of temporaries."""
    import torch

    if out is None:
        out = torch.empty(count, dtype=torch.float32, device=device)
    period = int(round(PERIOD_SECONDS * sample_rate))
    pulse = int(round(PULSE_SECONDS * sample_rate))
    k = (PULSE_F1 - PULSE_F0) / PULSE_SECONDS
    key = (int(seed) * 0x9E3779B1 + int(stream) * 0x85EBCA77) & 0xFFFFFFFF

    def mix32(v):
        v = v & 0xFFFFFFFF
        v = v ^ (v >> 16)
        v = (v * 0x7FEB352D) & 0xFFFFFFFF
        v = v ^ (v >> 15)
        v = (v * 0x846CA68B) & 0xFFFFFFFF
        v = v ^ (v >> 16)
        return v

    done = 0
    while done < count:
        m = min(chunk, count - done)
        i = torch.arange(start + done, start + done + m, dtype=torch.int64, device=out.device)
        r = i % period
        tt = r.to(torch.float64) / sample_rate
        phase = 2.0 * np.pi * (PULSE_F0 * tt + 0.5 * k * tt * tt)
        sig = torch.where(r < pulse, AMPLITUDE * torch.sin(phase), torch.zeros_like(phase))
        h = mix32((i & 0xFFFFFFFF) ^ key)
        h = mix32(h + ((i >> 32) & 0xFFFFFFFF))
        u = (h >> 8).to(torch.float64) / float(1 << 24)
        out[done:done + m] = (sig + NOISE * (u - 0.5)).to(torch.float32)
        done += m
    return out
