"""bench.py --gpus N without a launcher around it starts N ranks itself (CPU checks of that path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sono_bench", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_spawns_n_ranks_of_itself(monkeypatch):
    b = load_bench()
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setenv("SONO_BENCH_BACKEND", "gloo")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert b.main() == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]      # the ranks see the same arguments


def fake_topology(tmp_path, simd_counts):
    for i, simds in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simds else 64}\nsimd_count {simds}\nmem_banks_count 1\n")
    return str(tmp_path)


def test_gpus_are_counted_from_the_kfd_topology_without_hip(tmp_path, monkeypatch):
    b = load_bench()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    topo = fake_topology(tmp_path, [0, 0, 1024, 1024, 1024])        # two CPU nodes, three GPUs
    assert b.visible_gpus(topo) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert b.visible_gpus(topo) == 2
    assert b.visible_gpus(str(tmp_path / "missing")) is None
    assert "torch" not in b.visible_gpus.__code__.co_names          # no torch, hence no HIP, in the launcher's count


def test_too_few_gpus_fails_loudly(tmp_path, monkeypatch):
    b = load_bench()
    topo = fake_topology(tmp_path, [0, 1024, 1024])
    monkeypatch.setattr(b, "visible_gpus", lambda topology=topo, f=b.visible_gpus: f(topology))
    monkeypatch.setattr(subprocess, "call", lambda cmd: pytest.fail("ranks were started"))
    monkeypatch.delenv("SONO_BENCH_BACKEND", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as ei:
        b.main()
    assert "needs 4 visible GPUs" in str(ei.value) and "shows 2" in str(ei.value)


def test_unreadable_topology_starts_the_ranks_anyway(monkeypatch):
    b = load_bench()
    seen = {}
    monkeypatch.setattr(b, "visible_gpus", lambda: None)
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.delenv("SONO_BENCH_BACKEND", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    assert b.main() == 0 and "--nproc-per-node=2" in seen["cmd"]


def test_committed_counters_are_quoted_only_for_the_build_they_were_taken_on(tmp_path):
    import json
    b = load_bench()
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"one build of the library")
    h = b.library_sha256(str(lib))
    assert h and len(h) == 64 and b.library_sha256(str(tmp_path / "none.so")) is None
    prof = tmp_path / "profiles"
    prof.mkdir()
    traffic = {"fft_size": b.N_FFT, "noverlap": b.NOVERLAP, "seconds": b.SECONDS, "hbm_bytes": 8.6e9, "source": "x", "lib_sha256": h}
    (prof / "hbm_traffic.json").write_text(json.dumps(traffic))
    (prof / "valu_counters.json").write_text(json.dumps({"insts_valu": 1.7e9, "lib_sha256": h}))
    t, note = b.measured_traffic(h, str(prof))
    assert t["hbm_bytes"] == 8.6e9 and note is None
    v, note = b.measured_valu(h, str(prof))
    assert v["insts_valu"] == 1.7e9 and note is None
    # a rebuilt library: the same files must not be quoted any more, and the line says why
    lib.write_bytes(b"another build")
    h2 = b.library_sha256(str(lib))
    t, note = b.measured_traffic(h2, str(prof))
    assert t is None and "another build" in note
    v, note = b.measured_valu(h2, str(prof))
    assert v is None and "another build" in note
    # counters from before the hash was recorded, another workload, no file at all
    (prof / "valu_counters.json").write_text(json.dumps({"insts_valu": 1.7e9}))
    assert b.measured_valu(h, str(prof))[0] is None
    traffic["seconds"] = 60
    (prof / "hbm_traffic.json").write_text(json.dumps(traffic))
    assert b.measured_traffic(h, str(prof))[0] is None
    assert b.measured_traffic(h, str(tmp_path / "empty"))[0] is None


def test_world_size_must_match_gpus(monkeypatch):
    b = load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as ei:
        b.main()
    assert "WORLD_SIZE=2" in str(ei.value)
