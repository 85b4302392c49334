"""GPU: the CHECKED build of the library (make -C moving_object_detector_amd/csrc checked) — every index the cluster kernels derive
from data in memory is verified before use, a violation is counted instead of faulting the GPU — runs the shapes of the one GPU
memory fault this repository has seen (a 64-pair 1280 x 720 batch under the profiler, on a build between
two commits) and the other index-critical shapes; every counter must stay 0 and the results must equal the oracle's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_no_index_violation_on_the_critical_shapes():
    lib = os.path.join(ROOT, "moving_object_detector_amd", "libmod_sf_checked.so")
    if not os.path.exists(lib):                                # normally built by __graft_entry__.build(); hipcc is on the GPU box too
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "moving_object_detector_amd", "csrc"), "checked"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checked_worker.py")], env=dict(os.environ, MOD_SF_LIB=lib),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(out["ran"]) == 7
    assert all(v == 0 for v in out["violations"].values()), out["violations"]
