"""CPU: the C-ABI library loads, exports every symbol include/mod_sf.h declares, mirrors its struct layouts, and fails
loudly — never falls back — when there is no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mod_sf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|void|const char \*)\s*\**\s*(mod_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return [n for n in names if n != "mod_mask_words"]


def test_header_and_binding_list_agree():
    from moving_object_detector_amd import capi
    assert sorted(declared_functions()) == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    from moving_object_detector_amd import capi
    lib = capi.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.mod_abi_version() == 2
    # diagnostic entry points (csrc/mod_sf_debug.h) exist only in PHASE_COUNTERS / ABLATE builds
    assert not hasattr(lib, "mod_debug_read") and not hasattr(lib, "mod_debug_counters")


def test_library_exports_the_c_abi_and_nothing_else():
    """A drop-in C library exports its C ABI only: `nm -D --defined-only` lists the declarations of include/mod_sf.h — no C++ launcher,
    no template instantiation, no compiler-emitted global (csrc/Makefile: -fvisibility=hidden + csrc/exports.map)."""
    import subprocess
    from moving_object_detector_amd import capi
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert names == sorted(declared_functions()), set(names) ^ set(declared_functions())
    checked = os.path.join(os.path.dirname(capi.LIB_PATH), "libmod_sf_checked.so")
    if os.path.exists(checked):                      # the diagnostic twin adds csrc/mod_sf_debug.h's two entry points
        out = subprocess.run(["nm", "-D", "--defined-only", checked], capture_output=True, text=True, check=True).stdout
        extra = sorted(set(line.split()[-1] for line in out.splitlines() if line.strip()) - set(names))
        assert extra == ["mod_debug_counters", "mod_debug_read"], extra


def test_struct_layouts():
    from moving_object_detector_amd import capi
    assert C.sizeof(capi.ModObject) == 112
    assert C.sizeof(capi.ModCamera) == 72 and C.sizeof(capi.ModParams) == 32 and C.sizeof(capi.ModTransform) == 56
    assert capi.ModObject.center.offset == 8 and capi.ModObject.velocity.offset == 64 and capi.ModObject.bounding_box.offset == 88


def test_no_device_is_an_error_not_a_fallback():
    import torch
    from moving_object_detector_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load()
    cfg = capi.ModConfig(0, 64, 48, 1, 0, 0, None)
    h = C.c_void_p()
    assert lib.mod_create(C.byref(cfg), C.byref(h)) == capi.MOD_ERR_NO_DEVICE
    assert not h.value
    from moving_object_detector_amd.pipeline import Context
    with pytest.raises(RuntimeError):
        Context(64, 48)


def test_config_limits_are_checked_before_any_device_work():
    from moving_object_detector_amd import capi
    lib = capi.load()
    h = C.c_void_p()
    for cfg in (capi.ModConfig(0, 64, 48, 70000, 0, 0, None),        # frames ride in grid.y / grid.z
                capi.ModConfig(0, 16384, 8192, 1, 0, 0, None),       # 2^27 px: 32-bit byte offsets of the AoS cloud
                capi.ModConfig(0, 64, 48, 65535, 40000, 0, None),    # frame * max_objects + cluster must fit 31 bits
                capi.ModConfig(0, 16385, 16, 1, 0, 0, None),         # wider than MOD_MAX_WIDTH
                capi.ModConfig(0, 0, 48, 1, 0, 0, None)):
        assert lib.mod_create(C.byref(cfg), C.byref(h)) == capi.MOD_ERR_INVALID_ARGUMENT
        assert not h.value


def test_missing_library_raises(monkeypatch, tmp_path):
    from moving_object_detector_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libmod_sf.so"))
    with pytest.raises(ImportError):
        capi.load()


def test_product_does_not_reach_the_oracle():
    """The oracle is test infrastructure: nothing in the package imports it, the library does not link it, and bench.py
    touches it only inside its cpu_baseline leg."""
    import ast
    import re
    import subprocess
    pkg = os.path.join(ROOT, "moving_object_detector_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                tree = ast.parse(open(os.path.join(dirpath, fn)).read())
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or ""]
                    assert not any(n.split(".")[0] == "oracle" for n in names), (fn, names)
            if fn.endswith((".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'#include\s+["<][^">]*oracle', src), fn
    from moving_object_detector_amd import capi
    needed = subprocess.run(["readelf", "-d", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in needed and "lookup_ref" not in needed
    bench = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in ast.walk(bench):
        if isinstance(fn, ast.FunctionDef):
            uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in ast.walk(fn))
            assert not uses or fn.name == "cpu_baseline", fn.name


def test_only_k_final_touches_the_optional_labels_plane():
    """The cluster-label plane is optional (NULL = not wanted).  The GPU abort of 2026-10-04 10:53 was a kernel
    other than k_final — the tie replay's image-scan fallback — still reading it.  Source-level pin: `a.labels` appears in
    k_final only, and every store there sits behind a test of the pointer."""
    import re
    src = open(os.path.join(ROOT, "moving_object_detector_amd", "csrc", "cluster.hip")).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"__global__[^\n]*\bvoid\s+(\w+)\s*\(", src)]
    uses = [m.start() for m in re.finditer(r"\ba\.labels\b", src)]
    assert uses, "k_final is expected to write the labels plane"
    for u in uses:
        owner = [name for pos, name in starts if pos < u][-1]
        assert owner == "k_final", f"a.labels used in {owner}"
    lines = src.splitlines()
    for i, line in enumerate(lines):
        if re.search(r"\ba\.labels\s*\[", line):                     # an access: guarded on this line or by the enclosing `if (a.labels)`
            ctx = " ".join(lines[max(0, i - 2): i + 1])
            assert re.search(r"if\s*\(\s*a\.labels\b", ctx), f"cluster.hip:{i + 1}: unguarded access to the labels plane"


def test_product_library_has_no_hidden_switches_or_experiment_kernels():
    """A drop-in library does not change behaviour by environment: the product build reads no environment variable and carries no
    experiment kernel (round 3 shipped five getenv knobs, the slower one-wave tile kernel and an 8-px scene-flow variant)."""
    from moving_object_detector_amd import capi
    blob = open(os.path.join(ROOT, "moving_object_detector_amd", "libmod_sf.so"), "rb").read()
    for name in (b"MOD_TILE_KERNEL", b"MOD_SGM_PATH", b"MOD_SGM_GROUP", b"MOD_SGM_CU_KEEP", b"MOD_DEBUG", b"k_ccl_rows", b"k_scene_flow_v8"):
        assert name not in blob, name
    srcs = [os.path.join(ROOT, "moving_object_detector_amd", "csrc", f) for f in ("mod_sf.hip", "cluster.hip", "sceneflow.hip", "sgm.hip")]
    for path in srcs:
        for i, line in enumerate(open(path), 1):
            if "getenv" in line:
                # the one permitted use sits behind the diagnostic-build macros
                assert "MOD_DEBUG" in line and path.endswith("mod_sf.hip"), (path, i)
