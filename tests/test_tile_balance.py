"""Host logic that sizes the tiled kernel's per-XCD ranges (tile_balance_plan, csrc/sono_tile.hip), driven through the
library's test hook with hand-made workgroup records: no GPU needed."""
import ctypes

import numpy as np

import spectrogram_amd as sa

BLOCKS, WPB, TILES = 256, 16, 1_349_999      # the 1 h benchmark launch: 256 workgroups of 16 waves


def plan(records, share, stamp, seen, blocks=BLOCKS, resident=BLOCKS, tiles=TILES):
    fn = sa.lib().sono_debug_tile_balance
    fn.restype = ctypes.c_uint
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rec = np.ascontiguousarray(records, dtype=np.uint64)
    sh = np.array(share, dtype=np.float64)
    st, se = np.array([stamp], dtype=np.uint64), np.array([seen], dtype=np.uint64)
    pb = np.zeros(9, dtype=np.uint32)
    parts = fn(rec.ctypes.data, blocks, resident, WPB, tiles, sh.ctypes.data, st.ctypes.data, se.ctypes.data, pb.ctypes.data)
    return parts, sh, int(st[0]), int(se[0]), pb


def records(times_us, tiles_per_part, stamp, blocks=BLOCKS):
    """One record per workgroup: all start at tick 1000, workgroups of part p end after times_us[p]."""
    r = np.zeros((blocks, 4), dtype=np.uint64)
    for g in range(blocks):
        p = g % 8
        r[g] = (1000, 1000 + int(times_us[p] * 100), tiles_per_part[p], stamp)
    return r


def check_cut(pb, tiles=TILES):
    assert pb[0] == 0 and pb[8] == tiles
    assert all(pb[i] <= pb[i + 1] for i in range(8))


def test_first_launch_gets_equal_ranges_and_a_stamp():
    parts, share, stamp, seen, pb = plan(np.zeros((BLOCKS, 4)), [1.0] * 8, 0, 0)
    assert parts == 8 and stamp == 1 and seen == 0
    check_cut(pb)
    sizes = np.diff(pb.astype(np.int64))
    assert sizes.max() - sizes.min() <= 1 and np.allclose(share, 1.0)


def test_a_slow_part_gets_a_shorter_range():
    equal = [TILES // 8] * 8
    times = [2700, 2850, 2650, 2800, 2700, 2850, 2750, 2890]          # what the XCDs of one chip showed
    parts, share, stamp, seen, pb = plan(records(times, equal, 1), [1.0] * 8, 1, 0)
    assert parts == 8 and seen == 1 and stamp == 2
    check_cut(pb)
    sizes = np.diff(pb.astype(np.int64))
    assert sizes[2] == sizes.max() and sizes[7] == sizes.min()
    assert abs(share.sum() - 8.0) < 1e-9
    # half way towards the measured speeds
    speed = 1.0 / np.array(times, dtype=np.float64)
    want = 0.5 + 0.5 * speed * 8 / speed.sum()
    assert np.allclose(share, want * 8 / want.sum(), atol=1e-9)
    # repeated with times that follow the new cut, the shares settle where all parts take equally long
    for it in range(12):
        t = [sizes[p] * times[p] / equal[p] for p in range(8)]          # every part keeps its time per tile
        parts, share, stamp, seen, pb = plan(records(t, sizes, stamp), share, stamp, seen)
        sizes = np.diff(pb.astype(np.int64))
    t = np.array([sizes[p] * times[p] / equal[p] for p in range(8)])
    assert t.max() / t.min() < 1.002


def test_reports_are_taken_up_once_and_only_whole():
    equal = [TILES // 8] * 8
    times = [2700, 2850, 2650, 2800, 2700, 2850, 2750, 2890]
    rec = records(times, equal, 3)
    # already taken up (seen == its stamp): the cut stays
    parts, share, stamp, seen, pb = plan(rec, [1.0] * 8, 3, 3)
    assert parts == 8 and np.allclose(share, 1.0) and seen == 3 and stamp == 4
    # a launch still running: some workgroups carry the older stamp
    torn = rec.copy()
    torn[100:, 3] = 2
    parts, share, stamp, seen, pb = plan(torn, [1.0] * 8, 3, 1)
    assert parts == 8 and np.allclose(share, 1.0) and seen == 1
    # a stamp no launch of this engine was given, a record that ends before it starts: left alone
    parts, share, _, seen, _ = plan(records(times, equal, 9), [1.0] * 8, 3, 1)
    assert np.allclose(share, 1.0) and seen == 1
    bad = rec.copy()
    bad[5, 1] = 10
    parts, share, _, seen, _ = plan(bad, [1.0] * 8, 3, 1)
    assert np.allclose(share, 1.0) and seen == 1


def test_wild_reports_are_refused_and_shares_are_clamped():
    equal = [TILES // 8] * 8
    # one part ten times slower than the rest: not believable, nothing changes
    parts, share, _, seen, pb = plan(records([2700] * 7 + [27000], equal, 1), [1.0] * 8, 1, 0)
    assert np.allclose(share, 1.0) and seen == 1
    check_cut(pb)
    # a part 1.9 times slower, again and again: its range never falls below 0.8 of the mean before normalisation
    share, stamp, seen = np.ones(8), 1, 0
    for _ in range(10):
        parts, share, stamp, seen, pb = plan(records([2700] * 7 + [5100], equal, stamp), share, stamp, seen)
        check_cut(pb)
    assert share.min() > 0.75 and abs(share.sum() - 8.0) < 1e-9


def test_small_or_uneven_launches_keep_equal_ranges():
    z = np.zeros((BLOCKS, 4))
    assert plan(z, [1.0] * 8, 0, 0, tiles=BLOCKS * WPB * 64 - 1)[0] == 0          # fewer than 64 tiles per wave
    assert plan(z, [1.0] * 8, 0, 0, blocks=100, resident=256)[0] == 0             # the chip is not full
    assert plan(z, [1.0] * 8, 0, 0, blocks=252, resident=252)[0] == 0             # workgroups do not divide among 8 parts
    assert plan(z, [1.0] * 8, 0, 0, tiles=1 << 32)[0] == 0                        # tile numbers beyond 32 bits
    parts, _, stamp, _, _ = plan(z, [1.0] * 8, 7, 0, tiles=1000)
    assert parts == 0 and stamp == 7                                             # no stamp is spent on such a launch


def test_fixed_tiles_switch(monkeypatch):
    monkeypatch.setenv("SONO_FIXED_TILES", "1")
    assert plan(np.zeros((BLOCKS, 4)), [1.0] * 8, 0, 0)[0] == 0
