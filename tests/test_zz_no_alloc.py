"""sono_detect_runs_static_dev (the entry point that allocates nothing) against
sono_detect_runs_dev, through the C ABI: same run lists, bit for bit, on two calls
in a row, the second with a smaller input (the fixed device scratch keeps the first
call's contents), plus the edge shapes of the flag array and the size limit."""
import ctypes as C

import numpy as np
import pytest

import spectrogram_amd as sa

pytestmark = pytest.mark.gpu

MAX_FRAMES = 67108864          # SONO_RUNS_STATIC_MAX_FRAMES


def numpy_runs(active):
    a = np.concatenate([[0], (active != 0).astype(np.int8), [0]])
    d = np.diff(a)
    return np.flatnonzero(d == 1).astype(np.uint64), (np.flatnonzero(d == -1) - 1).astype(np.uint64)


def run_both(e, active):
    import torch
    frames = len(active)
    d_active = torch.from_numpy(np.ascontiguousarray(active, dtype=np.uint8)).cuda()
    out = {}
    for name in ("sono_detect_runs_dev", "sono_detect_runs_static_dev"):
        first = torch.full((frames // 2 + 1,), 0xDEAD, dtype=torch.int64, device="cuda")
        last = torch.full((frames // 2 + 1,), 0xBEEF, dtype=torch.int64, device="cuda")
        count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        fn = getattr(sa.lib(), name)
        sa._check(fn(e._handle(), d_active.data_ptr(), frames, first.data_ptr(), last.data_ptr(), count.data_ptr(), None))
        sa._check(sa.lib().sono_engine_synchronize(e._handle()))
        n = int(count.item())
        out[name] = (n, first[:n].cpu().numpy().astype(np.uint64), last[:n].cpu().numpy().astype(np.uint64),
                     first[n:].cpu().numpy(), last[n:].cpu().numpy())
    return out["sono_detect_runs_dev"], out["sono_detect_runs_static_dev"]


def check(e, active):
    ref, got = run_both(e, active)
    f, l = numpy_runs(active)
    assert got[0] == ref[0] == len(f)
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    assert np.array_equal(got[1], f) and np.array_equal(got[2], l)
    # nothing is written past the run count
    assert (got[3] == 0xDEAD).all() and (got[4] == 0xBEEF).all()


def test_two_calls_in_a_row_second_smaller():
    rng = np.random.default_rng(7)
    e = sa.SpectrogramEngine(1024, "hann")
    big = (rng.random(3_000_001) < 0.3).astype(np.uint8)           # 733 blocks of 4096 flags, the last one partial
    small = (rng.random(70_001) < 0.6).astype(np.uint8)            # 18 blocks: most of the scratch keeps the first call's sums
    check(e, big)
    check(e, small)
    check(e, big[:4096])                                           # exactly one block
    e.free()


@pytest.mark.parametrize("pattern", ["zeros", "ones", "alternating", "edges", "one_flag", "block_boundary"])
def test_edge_shapes(pattern):
    n = 3 * 4096 + 5
    a = np.zeros(n, np.uint8)
    if pattern == "ones":
        a[:] = 1
    elif pattern == "alternating":
        a[::2] = 1
    elif pattern == "edges":
        a[0] = a[-1] = 1
    elif pattern == "one_flag":
        a[4095] = 7                                                # any non-zero byte is a flag
    elif pattern == "block_boundary":
        a[4090:4100] = 1                                           # a run across two blocks
        a[8191] = 1
        a[8192] = 1
        a[12287:12289] = 1
    e = sa.SpectrogramEngine(256, "hann")
    check(e, a)
    for m in (1, 2, 15, 16, 17, 4095, 4097):
        check(e, a[:m])
        check(e, np.ones(m, np.uint8))
    e.free()


def test_empty_and_limit():
    import torch
    e = sa.SpectrogramEngine(512, "hann")
    count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    fn = sa.lib().sono_detect_runs_static_dev
    sa._check(fn(e._handle(), None, 0, None, None, count.data_ptr(), None))
    sa._check(sa.lib().sono_engine_synchronize(e._handle()))
    assert int(count.item()) == 0
    dummy = torch.zeros(16, dtype=torch.int64, device="cuda")
    rc = fn(e._handle(), dummy.data_ptr(), MAX_FRAMES + 1, dummy.data_ptr(), dummy.data_ptr(), count.data_ptr(), None)
    assert rc == 1                                                 # SONO_EINVAL, before anything is launched
    assert b"SONO_RUNS_STATIC_MAX_FRAMES" in sa.lib().sono_last_error()
    e.free()
