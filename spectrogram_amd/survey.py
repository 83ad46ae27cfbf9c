"""Batch detection over many recordings / channels, sharded over the GPUs of a node: what
`runAutoDetection` does for the one file open in the browser (wsManager.js:241-299: the
analysis engine SpectrogramEngine(1024, 'hann'), `processFullFile` on channel 0, then the
call-summary table, callSummaryTable.js:103-137) done for a list of units, one process per
GPU.  Units are dealt longest-first to the least-loaded rank when their lengths are known
(sharding.deal_units; round-robin otherwise), every rank runs the whole detector
(sono_process_full_file: fast scan, regions, framing, measurement) on its units, a unit that
fails is set aside without stopping the rank, and the per-rank tables are gathered with the
path's only collective.

`sharded_peaks` is the other partition: ONE long recording cut into frame chunks with an
fft_size - hop halo, the peak threshold of get_peaks (lib.rs:349-387) taken from the
recording's global maximum, which is one all-reduce(max) of a float.
"""
import numpy as np

from . import SpectrogramEngine, frames_for, framing, sharding

__all__ = ["detect_units", "run_auto_detection", "units_of_rank", "sharded_peaks", "ANALYSIS_FFT", "ANALYSIS_WINDOW",
           "REJECT_REFERENCE_THROWS", "REJECT_LOAD_FAILED", "REJECT_DETECTION_FAILED"]

ANALYSIS_FFT, ANALYSIS_WINDOW = 1024, "hann"          # getAnalysisWasmEngine, wsManager.js:350-361


# Why a unit contributed no rows (third column of the rejected table)
REJECT_REFERENCE_THROWS = 1      # the reference itself raises on this unit outside its debug mode (measure.ReferenceThrows)
REJECT_LOAD_FAILED = 2           # the unit's samples could not be produced (its loader raised, or the array is unusable)
REJECT_DETECTION_FAILED = 3      # the library reported an error on this unit (bad arguments, out of memory, a HIP error)


def _samples(unit):
    """A unit's samples: float32, or int16 as they are (a decoded 16-bit WAV channel; the library takes those without a
    float copy on the host, framing._full_file_call_array)."""
    x = unit() if callable(unit) else unit
    if isinstance(x, np.ndarray) and x.dtype == np.int16:
        return np.ascontiguousarray(x)
    return np.ascontiguousarray(x, dtype=np.float32)


def _unit_length(u):
    """Samples of a unit when known without loading it: the array's size, or the optional fourth field."""
    if len(u) > 3 and u[3] is not None:
        return int(u[3])
    return None if callable(u[2]) else int(np.size(u[2]))


def units_of_rank(units, rank, world):
    """Indices of this rank's units: longest first to the least-loaded rank (sharding.deal_units) when every unit's
    length is known -- arrays, or loaders with a length as fourth field -- else round-robin (sharding.files_of_rank)."""
    lengths = [_unit_length(u) for u in units]
    if world > 1 and all(n is not None for n in lengths):
        return sharding.deal_units(lengths, world)[rank]
    return sharding.files_of_rank(len(units), rank, world)


def _shard_path(shard_dir, key, file_id, channel):
    import os
    return os.path.join(shard_dir, f"unit_{key}_{int(file_id)}_{int(channel)}.npz")


def _settings_key(sample_rate, flow_khz, fhigh_khz, options, config):
    import hashlib
    import json
    text = json.dumps([float(sample_rate), float(flow_khz), float(fhigh_khz), options or {}, config or {}], sort_keys=True, default=str)
    return hashlib.sha256(text.encode()).hexdigest()[:12]


def detect_units(units, sample_rate, flow_khz=10.0, fhigh_khz=128.0, options=None, config=None, rank=0, world=1,
                 engine=None, threads=0, shard_dir=None, detect=None, in_flight=1):
    """This rank's share of `units` = [(file_id, channel, samples or a callable returning them[, length])]: the
    reference's whole-file detection on each, calls as float64 rows in sharding.COLUMNS order (file, channel, then the
    table's numeric columns).  Units keep their order within the rank.

    A unit that fails is CONTAINED: it contributes no rows, is named in the second return value as
    (file_id, channel, REJECT_* code, message), and the rank goes on with its next unit -- one bad file or one unit that
    does not fit must not cost the night's other recordings, and above all must not keep this rank out of the gather
    the other ranks are waiting in.

    shard_dir: every finished unit leaves `unit_<settings>_<file>_<channel>.npz` there (rows and status, written to a
    temporary name and renamed), and a unit whose shard exists is read back instead of detected again: a survey that
    was killed is started again with the same arguments and skips what it had finished.

    in_flight: units worked on side by side on this rank's GPU, each on a thread and an engine of its own (the library
    calls leave the interpreter lock).  A short recording's detection is a chain of small launches and waits -- 3 ms
    for five seconds of audio, of which the GPU is busy for a fraction -- so a survey of many short files gains from
    several in flight; hour-long recordings fill the card and the link on their own.  Same table, same order.

    detect: the per-unit detector, (engine, samples, file_id, channel) -> rows; default the library's whole-file
    detection.  (The CPU tests of the sharding logic pass a stand-in.)"""
    import os
    import threading
    from . import SonoError, measure as _ms
    own = engine is None and detect is None
    default_detect = detect is None
    if detect is None:
        def detect(eng, x, file_id, channel):
            return framing.process_full_file_table(eng, x, sample_rate, flow_khz, fhigh_khz, options, config, threads=threads,
                                                   file_id=file_id, channel=channel)
    key = _settings_key(sample_rate, flow_khz, fhigh_khz, options, config)
    if shard_dir:
        os.makedirs(shard_dir, exist_ok=True)
    empty = np.zeros((0, len(sharding.COLUMNS)), np.float64)
    mine = list(units_of_rank(units, rank, world))

    def one_unit(e, i):
        """-> (code, message, rows) of unit i, its shard written or read."""
        file_id, channel, unit = units[i][0], units[i][1], units[i][2]
        path = _shard_path(shard_dir, key, file_id, channel) if shard_dir else None
        if path and os.path.exists(path):
            with np.load(path, allow_pickle=False) as z:
                return int(z["code"]), str(z["message"]), z["rows"]
        code, msg, r = 0, "", empty
        try:
            x = _samples(unit)
        except Exception as err:          # noqa: BLE001 -- whatever a caller's loader raises stays with this unit
            code, msg = REJECT_LOAD_FAILED, f"{type(err).__name__}: {err}"
        if not code:
            try:
                r = np.asarray(detect(e, x, file_id, channel), np.float64).reshape(-1, len(sharding.COLUMNS))
            except _ms.ReferenceThrows as err:
                code, msg = REJECT_REFERENCE_THROWS, str(err)
            except (SonoError, MemoryError, ValueError, RuntimeError) as err:
                code, msg = REJECT_DETECTION_FAILED, f"{type(err).__name__}: {err}"
        if code:
            r = empty
        if path:
            tmp = path + f".tmp{os.getpid()}_{threading.get_ident()}"
            with open(tmp, "wb") as fh:
                np.savez(fh, rows=r, code=np.int64(code), message=np.str_(msg))
            os.replace(tmp, path)
        return code, msg, r

    workers = max(1, min(int(in_flight), len(mine)))
    if engine is not None and default_detect:
        workers = 1                           # one engine, one call at a time
    results = [None] * len(mine)
    if workers == 1:
        e = SpectrogramEngine(ANALYSIS_FFT, ANALYSIS_WINDOW) if own else engine
        try:
            for k, i in enumerate(mine):
                results[k] = one_unit(e, i)
        finally:
            if own and e is not None:
                e.free()
    else:
        # the threads draw the rank's units in order; what cannot be contained in a unit (an engine that cannot be
        # created, an exception outside the ones above) ends the survey on this rank like it does with one thread
        lock, state = threading.Lock(), {"next": 0, "fatal": None}

        def work():
            e = None
            try:
                e = SpectrogramEngine(ANALYSIS_FFT, ANALYSIS_WINDOW) if default_detect else engine
                while True:
                    with lock:
                        k = state["next"]
                        if k >= len(mine) or state["fatal"] is not None:
                            return
                        state["next"] = k + 1
                    results[k] = one_unit(e, mine[k])
            except BaseException as err:      # noqa: BLE001 -- raised again on the calling thread below
                with lock:
                    if state["fatal"] is None:
                        state["fatal"] = err
            finally:
                if default_detect and e is not None:
                    e.free()

        pool = [threading.Thread(target=work) for _ in range(workers)]
        for t in pool:
            t.start()
        for t in pool:
            t.join()
        if state["fatal"] is not None:
            raise state["fatal"]
    rows, rejected = [], []
    for k, i in enumerate(mine):
        code, msg, r = results[k]
        if code:
            rejected.append((units[i][0], units[i][1], code, msg))
        else:
            rows.append(r)
    table = np.concatenate(rows, axis=0) if rows else empty
    return table, rejected


def run_auto_detection(units, sample_rate, flow_khz=10.0, fhigh_khz=128.0, options=None, config=None, engine=None,
                       collective_device=None, threads=0, shard_dir=None, detect=None, in_flight=1):
    """detect_units on this process's rank of the initialised torch.distributed group (or on everything when there
    is none), then the gathered call-summary table, identical on every rank and ordered by (file, channel, start), and
    the gathered list of rejected units [(file, channel, REJECT_* code)], identical on every rank and ordered alike
    (the messages stay on the rank that saw them: detect_units returns them).

    Failures: a failing UNIT is contained by detect_units.  A rank that cannot go on at all (no device, no engine)
    does not leave: it tells the others in one all_reduce that every rank performs before the gather, and then ALL
    ranks raise RuntimeError -- nobody waits in a collective for a rank that will not arrive.  (A rank that is killed
    outright is the launcher's business: torchrun ends the other ranks.)"""
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    rows, rejected, fatal = None, [], None
    try:
        rows, rejected = detect_units(units, sample_rate, flow_khz, fhigh_khz, options, config, rank, world, engine, threads,
                                      shard_dir, detect, in_flight)
    except Exception as err:                  # noqa: BLE001 -- reported to every rank below, then raised
        fatal = err
    bad_rank = sharding.agree_on_failure(fatal is not None, device=collective_device)
    if bad_rank >= 0:
        if fatal is not None:
            raise RuntimeError(f"survey: rank {rank} cannot continue: {type(fatal).__name__}: {fatal}") from fatal
        raise RuntimeError(f"survey: rank {bad_rank} reported that it cannot continue; no table was gathered")
    table = sharding.gather_call_tables(rows, device=collective_device)
    rej = np.array([(f, c, code) for f, c, code, _ in rejected], np.float64).reshape(-1, 3)
    rej = sharding.gather_rows(rej, 3, device=collective_device)
    if rej.shape[0]:
        rej = rej[np.lexsort((rej[:, 1], rej[:, 0]))]
    return table, [(int(f), int(c), int(code)) for f, c, code in rej]


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
