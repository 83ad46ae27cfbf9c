"""The composed sharded pipeline (spectrogram_amd.survey) on the GPU: units dealt to ranks, whole-file detection per
unit, call-summary rows gathered -- two gloo ranks on one card against one process, row for row; and one long
recording cut into frame chunks whose peak thresholds come from the all-reduced global maximum."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANK = os.path.join(ROOT, "tests", "survey_rank.py")


def run_world(world, out_dir, *extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, RANK, str(r), str(world), port, str(out_dir), *extra]) for r in range(world)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0] * world


def test_two_ranks_gather_the_single_process_table(tmp_path):
    import spectrogram_amd as sa
    from spectrogram_amd import sharding
    from tests.survey_rank import long_recording
    run_world(1, tmp_path)
    run_world(2, tmp_path)
    one = np.load(tmp_path / "table_w1_r0.npy")
    assert one.shape[1] == len(sharding.COLUMNS) and one.shape[0] >= 8             # the recordings do hold calls
    assert sorted(set(one[:, 0].astype(int))) == [0, 1, 2, 3, 4] and set(one[one[:, 0] == 3][:, 1].astype(int)) == {0, 1}
    for r in range(2):
        two = np.load(tmp_path / f"table_w2_r{r}.npy")
        assert np.array_equal(one, two, equal_nan=True)                             # row for row, on every rank
    # Partition C: the chunks' peaks, put together, are get_peaks of the whole recording
    x = long_recording()
    e = sa.SpectrogramEngine(1024, "hann")
    e.compute_spectrogram_u8(x, 768, 20.0, 80.0)
    want, gmax = e.get_peaks(0.4), e.get_global_max()
    e.free()
    parts = [np.load(tmp_path / f"peaks_w2_r{r}.npz") for r in range(2)]
    assert int(parts[0]["f0"]) == 0 and int(parts[0]["f1"]) == int(parts[1]["f0"]) and int(parts[1]["f1"]) == len(want)
    assert int(parts[0]["f1"]) % 16 == 0 and 0 < int(parts[0]["f1"]) < len(want)
    assert np.array_equal(np.concatenate([p["peaks"] for p in parts]), want)
    assert float(parts[0]["gmax"]) == float(parts[1]["gmax"]) == float(gmax)
    whole = np.load(tmp_path / "peaks_w1_r0.npz")
    assert np.array_equal(whole["peaks"], want)
    assert (want != 0xFFFF).any() and (want == 0xFFFF).any()


def test_a_failing_unit_is_contained_and_every_rank_learns_of_it(tmp_path):
    """One unit's loader raises on the rank that owns it.  That rank sets the unit aside, finishes its other units and
    enters the gather like the others: both ranks end with the table of the good units -- the one a run without the bad
    unit gives -- and the same list of rejected units."""
    from spectrogram_amd import survey
    run_world(1, tmp_path)
    run_world(2, tmp_path, "bad")
    good = np.load(tmp_path / "table_w1_r0.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"table_w2_r{r}_bad.npy"), good, equal_nan=True)
        rej = np.load(tmp_path / f"rejected_w2_r{r}.npy")
        assert rej.tolist() == [[7, 0, survey.REJECT_LOAD_FAILED]]


def test_short_recordings_in_flight_give_the_table_of_one_at_a_time():
    """Twenty-four recordings of two to five seconds through survey.detect_units one at a time and four at a time (a
    thread and an engine each): the same rows in the same order."""
    import time
    from spectrogram_amd import survey, synth
    sr = 384000
    units = [(i, 0, synth.chirp_numpy(0, int((2 + i % 4) * sr), sr, stream=i)) for i in range(24)]
    t0 = time.perf_counter()
    one, rej1 = survey.detect_units(units, sr, config={"debugMode": True})
    t1 = time.perf_counter()
    four, rej4 = survey.detect_units(units, sr, config={"debugMode": True}, in_flight=4)
    t2 = time.perf_counter()
    print(f"one at a time {1e3 * (t1 - t0):.1f} ms, four in flight {1e3 * (t2 - t1):.1f} ms, {one.shape[0]} calls")
    assert one.shape[0] > 24 * 15 and rej1 == rej4 == []
    assert np.array_equal(one.view(np.uint64), four.view(np.uint64))
