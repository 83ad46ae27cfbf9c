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
