"""GPU: the process-group code of the N-rank run executes on RCCL — a one-rank communicator on cuda:0, in a fresh child
(the pool's boxes have one GPU, so this is the part of BASELINE config 4's collective that can run here)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_runs_the_jobs_collectives():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["camera_roundtrip"] and out["params_roundtrip"]          # the struct block survives the device broadcast byte for byte
    assert out["max_over_ranks"] == 0.123456789                         # F64 through the device all-reduce, unchanged
    assert out["gathered"] == [[0, 1, 2, 3, 4]]
    assert any("rccl" in p for p in out["rccl_libraries_mapped"]), out  # librccl really is in the process


def test_bench_line_reports_the_collective_honestly(tmp_path):
    """bench.py at N = 1 goes through the same group code; the line says whether a collective ran, on what and with which backend."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--distinct", "4", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--no-config5", "--no-latency"]
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["config"]
    assert cfg["collective_executed"] is True and cfg["collective_backend"] == "nccl" and cfg["collective_on_device"] is True
    assert cfg["collective_world_size"] == 1 and cfg["collective_error"] is None
    r = subprocess.run(base + ["--no-dist"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["config"]
    assert cfg["collective_executed"] is False and cfg["collective"].startswith("none")
