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


def test_too_few_gpus_fails_loudly(monkeypatch):
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a machine with 64 GPUs")
    b = load_bench()
    monkeypatch.delenv("SONO_BENCH_BACKEND", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "64"])
    with pytest.raises(SystemExit) as ei:
        b.main()
    assert "needs 64 visible GPUs" in str(ei.value)


def test_world_size_must_match_gpus(monkeypatch):
    b = load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as ei:
        b.main()
    assert "WORLD_SIZE=2" in str(ei.value)
