"""The sharding layer: partition rules on one process, and the two collectives
on two gloo ranks (CPU).  No GPU compute: rows are synthetic."""
import os
import socket

import numpy as np
import pytest

from spectrogram_amd import sharding as sh


def test_files_of_rank_partitions_everything():
    for n, w in ((10, 1), (10, 3), (10000, 8), (3, 8)):
        got = sorted(i for r in range(w) for i in sh.files_of_rank(n, r, w))
        assert got == list(range(n))


def test_frame_chunks_cover_align_and_balance():
    for total, parts in ((5399997, 8), (9762, 2), (17, 4), (0, 3), (16, 8), (1123883, 8)):
        ch = sh.frame_chunks(total, parts)
        assert len(ch) == parts and ch[0][0] == 0 and ch[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all(f0 % sh.FRAME_ALIGN == 0 for f0, f1 in ch if f1 > f0)
        sizes = [f1 - f0 for f0, f1 in ch]
        assert max(sizes) - min(sizes) <= 2 * sh.FRAME_ALIGN


def test_chunk_samples_reproduce_the_frame_rule():
    N, H, n = 1024, 256, 1024 + 999 * 256 + 77
    total = (n - N) // H + 1
    for f0, f1 in sh.frame_chunks(total, 4):
        s0, s1 = sh.chunk_samples((f0, f1), N, H)
        assert s1 <= n and ((s1 - s0) - N) // H + 1 == f1 - f0          # lib.rs:118-123 on the slice gives exactly the chunk
    assert sh.chunk_samples((5, 5), N, H) == (5 * H, 5 * H)


def test_calls_to_rows_schema():
    calls = [dict(startTime_s=1.0, endTime_s=1.004, duration_ms=4.0, peakFreq_kHz=45.5, peakPower_dB=-30.0, peakTime_s=1.001)]
    rows = sh.calls_to_rows(calls, file_id=7, channel=1)
    col = {k: i for i, k in enumerate(sh.COLUMNS)}
    assert rows.shape == (1, len(sh.COLUMNS)) and rows[0, col["file"]] == 7 and rows[0, col["channel"]] == 1
    assert abs(rows[0, col["peakFreq_ms"]] - 1.0) < 1e-9 and np.isnan(rows[0, col["kneeFreq_kHz"]])
    assert sh.gather_call_tables(rows).shape == (1, len(sh.COLUMNS))      # no process group: identity


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank r owns files r, r+world, ...; file f contributes f % 3 rows (so some ranks may contribute none)
        rows = []
        for f in sh.files_of_rank(7, rank, world):
            for k in range(f % 3):
                r = np.full(len(sh.COLUMNS), np.nan)
                r[0], r[1], r[2], r[3] = f, 0, 10.0 - k, 10.5 - k
                rows.append(r)
        table = sh.gather_call_tables(np.array(rows).reshape(-1, len(sh.COLUMNS)))
        gmax = sh.reduce_global_max(0.25 * (rank + 1))
        np.save(os.path.join(out_dir, f"table{rank}.npy"), table)
        np.save(os.path.join(out_dir, f"gmax{rank}.npy"), np.array([gmax]))
    finally:
        dist.destroy_process_group()


def test_gather_call_tables_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tables = [np.load(tmp_path / f"table{r}.npy") for r in range(world)]
    want = sorted((f, 10.0 - k) for f in range(7) for k in range(f % 3))
    for t in tables:
        assert t.shape == (len(want), len(sh.COLUMNS))
        assert [(int(r[0]), r[2]) for r in t] == want               # every rank holds the same, ordered table
    assert np.array_equal(tables[0], tables[1], equal_nan=True)
    assert [float(np.load(tmp_path / f"gmax{r}.npy")[0]) for r in range(world)] == [0.5, 0.5]


# ---- the survey's robustness rules, on the CPU (a stand-in detector; the collectives are the real ones) --------------

def test_deal_units_longest_first():
    night = [12 * 3600] + [5] * 40                       # one continuous recording and forty 5 s triggers
    deal = sh.deal_units(night, 2)
    assert deal[0] == [0] and deal[1] == list(range(1, 41))            # round-robin would give rank 0 the long one AND half the rest
    deal = sh.deal_units([7, 3, 9, 1, 4, 6, 2, 8], 3)
    assert sorted(i for d in deal for i in d) == list(range(8))
    loads = [sum([7, 3, 9, 1, 4, 6, 2, 8][i] for i in d) for d in deal]
    assert max(loads) - min(loads) <= 3 and all(d == sorted(d) for d in deal)
    assert sh.deal_units([5] * 7, 3) == [sh.files_of_rank(7, r, 3) for r in range(3)]      # equal lengths: the round-robin deal
    assert sh.deal_units([], 4) == [[], [], [], []]


def _fake_units():
    rng = np.random.default_rng(5)
    units = []
    for f in range(7):
        n = 100 * (1 + (f * 3) % 5)
        units.append((f, 0, rng.standard_normal(n).astype(np.float32)))

    def unreadable():
        raise OSError("no such recording")
    units[4] = (4, 0, unreadable, 300)
    return units


def _fake_detect(engine, x, file_id, channel):
    """Stand-in for the whole-file detector: one row per 100 samples; file 2 makes the library fail."""
    import spectrogram_amd as sa
    if file_id == 2:
        raise sa.SonoError(-3, "out of device memory on this unit")
    rows = np.full((x.size // 100, len(sh.COLUMNS)), np.nan)
    rows[:, 0], rows[:, 1] = file_id, channel
    rows[:, 2] = np.arange(rows.shape[0]) * 0.01
    rows[:, 3] = float(np.abs(x).sum())
    return rows


def _survey_rank(rank, world, port, out_dir, mode):
    import torch.distributed as dist
    from spectrogram_amd import survey
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        detect = _fake_detect
        if mode == "fatal" and rank == 1:
            def detect(engine, x, file_id, channel):
                raise KeyError("this rank has lost its device")          # nothing detect_units contains
        try:
            table, rejected = survey.run_auto_detection(_fake_units(), 384000, detect=detect)
            np.save(os.path.join(out_dir, f"t{rank}.npy"), table)
            np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array(rejected, dtype=np.int64).reshape(-1, 3))
        except RuntimeError as err:
            with open(os.path.join(out_dir, f"err{rank}.txt"), "w") as fh:
                fh.write(str(err))
    finally:
        dist.destroy_process_group()


def _spawn(tmp_path, mode, world=2):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_survey_rank, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)


def test_failing_units_are_contained_and_all_ranks_agree(tmp_path):
    from spectrogram_amd import survey
    _spawn(tmp_path, "units")
    want, _ = survey.detect_units([u for u in _fake_units() if u[0] not in (2, 4)], 384000, detect=_fake_detect)
    want = want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))]
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"t{r}.npy"), want, equal_nan=True)
        assert np.load(tmp_path / f"r{r}.npy").tolist() == [[2, 0, survey.REJECT_DETECTION_FAILED], [4, 0, survey.REJECT_LOAD_FAILED]]
    # the messages stay with the rank that saw them
    _, rejected = survey.detect_units(_fake_units(), 384000, detect=_fake_detect)
    assert [(f, c, code) for f, c, code, _ in rejected] == [(2, 0, 3), (4, 0, 2)]
    assert "out of device memory" in rejected[0][3] and "no such recording" in rejected[1][3]


def test_a_rank_that_cannot_continue_fails_every_rank_instead_of_hanging(tmp_path):
    _spawn(tmp_path, "fatal")
    e0, e1 = (tmp_path / "err0.txt").read_text(), (tmp_path / "err1.txt").read_text()
    assert "rank 1 reported that it cannot continue" in e0
    assert "rank 1 cannot continue" in e1 and "lost its device" in e1
    assert not (tmp_path / "t0.npy").exists() and not (tmp_path / "t1.npy").exists()


def test_units_are_dealt_by_length_when_lengths_are_known():
    from spectrogram_amd import survey
    units = _fake_units()                                           # lengths 100, 400, 200, 500, (300), 100, 400
    assert survey.units_of_rank(units, 0, 2) == sh.deal_units([100, 400, 200, 500, 300, 100, 400], 2)[0]
    units[1] = (1, 0, lambda: np.zeros(10, np.float32))             # a loader without a length: round-robin
    assert survey.units_of_rank(units, 1, 2) == sh.files_of_rank(7, 1, 2)


def test_finished_units_are_read_back_from_their_shards(tmp_path):
    from spectrogram_amd import survey
    calls = []

    def counting(engine, x, file_id, channel):
        calls.append(file_id)
        return _fake_detect(engine, x, file_id, channel)
    first = survey.detect_units(_fake_units(), 384000, detect=counting, shard_dir=str(tmp_path))
    assert sorted(calls) == [0, 1, 2, 3, 5, 6]                      # 4 never got as far as the detector
    assert len(list(tmp_path.glob("unit_*.npz"))) == 7 and not list(tmp_path.glob("*.tmp*"))
    calls.clear()
    again = survey.detect_units(_fake_units(), 384000, detect=counting, shard_dir=str(tmp_path))
    assert calls == []                                              # a second run detects nothing again ...
    assert np.array_equal(first[0], again[0], equal_nan=True) and first[1] == again[1]      # ... and returns the same
    # other settings are other shards
    survey.detect_units(_fake_units()[:1], 384000, flow_khz=20.0, detect=counting, shard_dir=str(tmp_path))
    assert calls == [0]


def test_units_in_flight_give_the_table_of_one_at_a_time(tmp_path):
    """in_flight > 1: the rank's units on several threads, the same rows in the same order, failures contained and named
    alike, shards written alike; something no unit can contain ends the survey as it does with one thread."""
    import threading
    import time
    from spectrogram_amd import survey
    seen = set()

    def slow_first(engine, x, file_id, channel):                     # the first units finish last
        seen.add(threading.get_ident())
        time.sleep(0.05 if file_id < 2 else 0.0)
        return _fake_detect(engine, x, file_id, channel)
    one = survey.detect_units(_fake_units(), 384000, detect=_fake_detect)
    many = survey.detect_units(_fake_units(), 384000, detect=slow_first, in_flight=3, shard_dir=str(tmp_path))
    assert np.array_equal(one[0], many[0], equal_nan=True) and one[1] == many[1]
    assert len(seen) > 1 and len(list(tmp_path.glob("unit_*.npz"))) == 7 and not list(tmp_path.glob("*.tmp*"))

    def fatal(engine, x, file_id, channel):
        if file_id == 3:
            raise KeyError("this rank has lost its device")
        return _fake_detect(engine, x, file_id, channel)
    with pytest.raises(KeyError):
        survey.detect_units(_fake_units(), 384000, detect=fatal, in_flight=3)
