"""GPU: the alternative tile stage k_ccl_rows (one wave per tile, MOD_TILE_KERNEL=rows; neighbor_distance <= 4) gives the oracle's
labels and objects too — a short soak over random sizes / parameters / hostile inputs in a fresh process (the kernel choice is
read once per process), plus the parity case at BASELINE's size."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_rows_kernel_soak():
    env = dict(os.environ, MOD_TILE_KERNEL="rows")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "20", "7"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    used = [l for l in r.stdout.splitlines() if l.startswith("ok") and any(f" n={k} " in l for k in (1, 2, 3, 4))]
    assert len(used) >= 10, "the soak is expected to hit neighbor_distance <= 4 (the rows kernel) many times"


def test_rows_kernel_1280x720_defaults():
    env = dict(os.environ, MOD_TILE_KERNEL="rows")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-k",
                        "1280x720_reference_defaults or neighbor_distance_sweep"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
