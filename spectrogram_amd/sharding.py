"""Spreading independent recordings, channels or time chunks over the GPUs of a
node: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI
on ROCm, "gloo" on CPU), no collective on the data path.  The STFT of a unit
needs nothing from any other unit; the only exchanges are

  * the per-GPU call-summary tables, gathered at the end (row counts first, then
    the rows padded to the largest count), and
  * one float -- the global maximum -- when ONE recording is cut into time chunks
    and its peak thresholds (lib.rs:355) must refer to the whole recording.

The reference has no counterpart: it is a single browser thread.
"""
import math

import numpy as np

__all__ = ["COLUMNS", "files_of_rank", "deal_units", "frame_chunks", "chunk_samples", "calls_to_rows", "gather_rows",
           "gather_call_tables", "reduce_global_max", "agree_on_failure", "FRAME_ALIGN"]

# Numeric columns of the reference's call-summary table (callSummaryTable.js:103-137), after the two
# keys that identify where a row came from.  Columns a call does not carry stay NaN.
COLUMNS = ["file", "channel", "startTime_s", "endTime_s", "duration_ms", "bandwidth_kHz", "startFreq_kHz", "startFreq_ms",
           "endFreq_kHz", "endFreq_ms", "highFreq_kHz", "highFreq_ms", "lowFreq_kHz", "lowFreq_ms", "peakFreq_kHz",
           "peakFreq_ms", "characteristicFreq_kHz", "characteristicFreq_ms", "kneeFreq_kHz", "kneeFreq_ms",
           "heelFreq_kHz", "heelFreq_ms", "peakPower_dB", "snr_dB"]

# Chunk boundaries fall on multiples of this many frames: the tiled kernel transforms frames in pairs
# (and up to 16 per wave), and a chunk that starts on such a boundary reproduces the unchunked rows bit for bit.
FRAME_ALIGN = 16


def files_of_rank(n_files, rank, world_size):
    """Partition A: recording i goes to GPU i mod world_size."""
    return list(range(rank, n_files, world_size))


def deal_units(lengths, world_size):
    """Partition A for units of known, different lengths (a night of 5 s triggers next to a 12 h continuous
    recording): longest first, each to the rank with the least work so far (ties: the lowest rank).  -> one list of
    unit indices per rank, each in increasing order.  Every rank computes the same deal from the same lengths.  With
    equal lengths this is the round-robin deal of files_of_rank."""
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        out[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(o) for o in out]


def frame_chunks(total_frames, parts, align=FRAME_ALIGN):
    """Partition B/C: `parts` contiguous frame ranges [f0, f1) covering [0, total_frames), boundaries on
    multiples of `align`, sizes within one `align` of each other."""
    units = math.ceil(total_frames / align)
    out = []
    for p in range(parts):
        u0, u1 = units * p // parts, units * (p + 1) // parts
        out.append((min(u0 * align, total_frames), min(u1 * align, total_frames)))
    return out


def chunk_samples(frame_range, fft_size, hop):
    """Sample slice [s0, s1) holding frames [f0, f1): the chunk plus its halo of fft_size - hop samples."""
    f0, f1 = frame_range
    if f1 <= f0:
        return (f0 * hop, f0 * hop)
    return (f0 * hop, (f1 - 1) * hop + fft_size)


def calls_to_rows(calls, file_id, channel=0):
    """Calls of framing.process_full_file (or detector.process_full_file_front_end) -> float64 rows in COLUMNS order."""
    rows = np.full((len(calls), len(COLUMNS)), np.nan, np.float64)
    col = {k: i for i, k in enumerate(COLUMNS)}
    for r, c in enumerate(calls):
        rows[r, col["file"]] = file_id
        rows[r, col["channel"]] = channel
        for k in COLUMNS[2:]:                       # measured calls carry every column; None (the reference's null) stays NaN
            if c.get(k) is not None:
                rows[r, col[k]] = c[k]
        if "peakFreq_ms" not in c:
            rows[r, col["peakFreq_ms"]] = (c["peakTime_s"] - c["startTime_s"]) * 1000
    return rows


def gather_rows(rows, width, device=None, group=None):
    """All ranks contribute `rows` (n_i x width float64); every rank gets the concatenation in rank order.  Two
    collectives: all_gather of the counts, all_gather of the rows padded to the largest count."""
    import torch
    import torch.distributed as dist

    rows = np.asarray(rows, np.float64).reshape(-1, width)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    count = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    pad = max(max(counts), 1)
    mine = torch.full((pad, width), float("nan"), dtype=torch.float64, device=dev)
    if rows.shape[0]:
        mine[: rows.shape[0]] = torch.from_numpy(rows).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return np.concatenate([p[:n].cpu().numpy() for p, n in zip(parts, counts)], axis=0)


def gather_call_tables(rows, device=None, group=None):
    """All ranks contribute `rows` (n_i x len(COLUMNS) float64); every rank gets the concatenation ordered by
    (file, channel, startTime_s)."""
    table = gather_rows(rows, len(COLUMNS), device, group)
    if table.shape[0]:
        order = np.lexsort((table[:, 2], table[:, 1], table[:, 0]))
        table = table[order]
    return table


def agree_on_failure(failed, device=None, group=None):
    """One all_reduce(MAX) of a flag: -> -1 when no rank reports `failed`, else the highest rank that does.  A rank that cannot go on calls this with failed=True INSTEAD of leaving: the others, who call it with
    False before they enter the gather, then learn of it and raise as well rather than wait in a collective for a
    rank that will never arrive."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0 if failed else -1
    dev = torch.device(device) if device is not None else torch.device("cpu")
    t = torch.tensor([dist.get_rank(group) if failed else -1], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def reduce_global_max(local_max, device=None, group=None):
    """Partition C: the recording's global maximum from the chunks' maxima (one float, all_reduce MAX)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(local_max)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    t = torch.tensor([float(local_max)], dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
