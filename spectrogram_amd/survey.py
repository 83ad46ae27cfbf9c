"""Batch detection over many recordings / channels, sharded over the GPUs of a node: what
`runAutoDetection` does for the one file open in the browser (wsManager.js:241-299: the
analysis engine SpectrogramEngine(1024, 'hann'), `processFullFile` on channel 0, then the
call-summary table, callSummaryTable.js:103-137) done for a list of units, one process per
GPU.  Units are dealt round-robin (sharding.files_of_rank), every rank runs the whole
detector (sono_process_full_file: fast scan, regions, framing, measurement) on its units,
and the per-rank tables are gathered with the path's only collective.

`sharded_peaks` is the other partition: ONE long recording cut into frame chunks with an
fft_size - hop halo, the peak threshold of get_peaks (lib.rs:349-387) taken from the
recording's global maximum, which is one all-reduce(max) of a float.
"""
import numpy as np

from . import SpectrogramEngine, frames_for, framing, sharding

__all__ = ["detect_units", "run_auto_detection", "sharded_peaks", "ANALYSIS_FFT", "ANALYSIS_WINDOW"]

ANALYSIS_FFT, ANALYSIS_WINDOW = 1024, "hann"          # getAnalysisWasmEngine, wsManager.js:350-361


def _samples(unit):
    x = unit() if callable(unit) else unit
    return np.ascontiguousarray(x, dtype=np.float32)


def detect_units(units, sample_rate, flow_khz=10.0, fhigh_khz=128.0, options=None, config=None, rank=0, world=1,
                 engine=None, threads=0):
    """This rank's share of `units` = [(file_id, channel, samples or a callable returning them)]: the reference's
    whole-file detection on each, calls as float64 rows in sharding.COLUMNS order (file, channel, then the table's
    numeric columns).  Units keep their order within the rank; a unit on which the reference itself would raise
    (measure.ReferenceThrows, outside its debug mode) contributes no rows and is named in the second return value."""
    from . import measure as _ms
    own = engine is None
    e = SpectrogramEngine(ANALYSIS_FFT, ANALYSIS_WINDOW) if own else engine
    rows, rejected = [], []
    try:
        for i in sharding.files_of_rank(len(units), rank, world):
            file_id, channel, unit = units[i]
            try:
                rows.append(framing.process_full_file_table(e, _samples(unit), sample_rate, flow_khz, fhigh_khz, options, config,
                                                            threads=threads, file_id=file_id, channel=channel))
            except _ms.ReferenceThrows:
                rejected.append((file_id, channel))
    finally:
        if own:
            e.free()
    table = np.concatenate(rows, axis=0) if rows else np.zeros((0, len(sharding.COLUMNS)), np.float64)
    return table, rejected


def run_auto_detection(units, sample_rate, flow_khz=10.0, fhigh_khz=128.0, options=None, config=None, engine=None,
                       collective_device=None, threads=0):
    """detect_units on this process's rank of the initialised torch.distributed group (or on everything when there
    is none), then the gathered call-summary table, identical on every rank and ordered by (file, channel, start)."""
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    rows, rejected = detect_units(units, sample_rate, flow_khz, fhigh_khz, options, config, rank, world, engine, threads)
    return sharding.gather_call_tables(rows, device=collective_device), rejected


def sharded_peaks(engine, audio, noverlap, threshold_ratio, rank=0, world=1, gain_db=20.0, range_db=80.0,
                  collective_device=None):
    """get_peaks(threshold_ratio) of a recording whose frames are cut into `world` chunks (sharding.frame_chunks, halo of
    fft_size - hop samples): this rank transforms its chunk only, the global maximum comes from one all-reduce, and the
    return value is (f0, f1, peaks of frames [f0, f1), global maximum).  Concatenated over the ranks the peaks equal
    get_peaks of the whole recording on one GPU: empty when the global maximum is <= 0 (lib.rs:350-352)."""
    x = np.ascontiguousarray(audio, dtype=np.float32)
    n_fft = engine.get_fft_size()
    hop = n_fft - noverlap
    total = frames_for(x.size, n_fft, noverlap)
    f0, f1 = sharding.frame_chunks(total, world)[rank]
    s0, s1 = sharding.chunk_samples((f0, f1), n_fft, hop)
    local_max, bins, mags = 0.0, np.zeros(0, np.uint16), np.zeros(0, np.float32)
    if f1 > f0:
        engine.compute_spectrogram_u8(x[s0:s1], noverlap, gain_db, range_db)
        local_max = float(engine.get_global_max())
        if local_max > 0.0:
            bins, mags = engine.get_peaks(0.0), engine.get_peak_magnitudes(0.0)
        else:                                            # a silent chunk: every frame's maximum is (bin 0, 0.0), lib.rs:370
            bins, mags = np.zeros(f1 - f0, np.uint16), np.zeros(f1 - f0, np.float32)
    gmax = np.float32(sharding.reduce_global_max(local_max, device=collective_device))
    if not gmax > 0.0:
        return f0, f1, np.zeros(0, np.uint16), float(gmax)
    threshold = np.float32(gmax * np.float32(threshold_ratio))                       # lib.rs:355
    peaks = np.where(mags >= threshold, bins, np.uint16(0xFFFF)).astype(np.uint16)   # lib.rs:380-384
    return f0, f1, peaks, float(gmax)
