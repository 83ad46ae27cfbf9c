"""One rank of the sharded survey (helper of tests/test_gpu_survey.py, started as a child process):
  python tests/survey_rank.py <rank> <world> <port> <out_dir>
Backend gloo, every rank on cuda:0: the multi-rank path on a one-GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_units(n_files=5, sr=384000, with_bad_units=False):
    """Small synthetic recordings with bat-pass-like FM pulses (spectrogram_amd.synth), two channels for file 3.
    with_bad_units: also a unit whose loader raises (file 7) and one the reference's own code raises on (file 8 would
    need such a recording; the contained-failure path is exercised with the loader)."""
    from spectrogram_amd import synth
    units = []
    for f in range(n_files):
        n = int(sr * (0.25 + 0.05 * f))
        units.append((f, 0, synth.chirp_numpy(0, n, sr, stream=f)))
        if f == 3:
            units.append((f, 1, synth.chirp_numpy(0, n, sr, stream=100 + f)))
    if with_bad_units:
        def unreadable():
            raise OSError("recording 7 is not there")
        units.insert(2, (7, 0, unreadable, int(sr * 0.3)))
    return units, sr


def long_recording(sr=384000):
    from spectrogram_amd import synth
    return synth.chirp_numpy(0, 1024 + 256 * 4001 + 17, sr, stream=9)


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch.distributed as dist
    import spectrogram_amd as sa
    from spectrogram_amd import survey
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bad = len(sys.argv) > 5 and sys.argv[5] == "bad"
        units, sr = make_units(with_bad_units=bad)
        table, rejected = survey.run_auto_detection(units, sr, 10.0, 128.0, config={"debugMode": True})
        np.save(os.path.join(out_dir, f"table_w{world}_r{rank}{'_bad' if bad else ''}.npy"), table)
        if bad:
            np.save(os.path.join(out_dir, f"rejected_w{world}_r{rank}.npy"), np.array(rejected, dtype=np.int64).reshape(-1, 3))
            return
        e = sa.SpectrogramEngine(1024, "hann")
        f0, f1, peaks, gmax = survey.sharded_peaks(e, long_recording(), 768, 0.4, rank, world)
        e.free()
        np.savez(os.path.join(out_dir, f"peaks_w{world}_r{rank}.npz"), f0=f0, f1=f1, peaks=peaks, gmax=gmax,
                 rejected=np.array(rejected, dtype=np.int64).reshape(-1, 3))
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
