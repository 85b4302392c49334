"""CPU: the product's replay of libstdc++ introsort (moving_object_detector_amd/csrc/introsort_emul.h, host build) against
the real std::sort / heap functions with the reference's comparator — 15 000 fuzzed checks incl. heavy ties."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_introsort_emulation_matches_libstdcxx(tmp_path):
    exe = str(tmp_path / "introsort_emul_test")
    subprocess.check_call(["g++", "-std=c++14", "-O2", os.path.join(ROOT, "tests", "cpp", "introsort_emul_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok")


def test_exact_div_known_reciprocal_matches_ieee_division(tmp_path):
    """csrc/exact_div.h div_by_known (velocity / dt through the host's 1/dt) against `/` on 40 M operand pairs."""
    exe = str(tmp_path / "exact_div_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "exact_div_test.cpp"),
                           "-o", exe])
    r = subprocess.run([exe, "100000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK")
