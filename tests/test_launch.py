"""CPU: `bench.py --gpus N` really starts N ranks (fresh children, torch.distributed.run) and a stream shards correctly."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks():
    r = _bench("--gpus", "2", "--launch-check", "--distinct", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["shards"] == [[0, 5], [5, 10]] and j["camera_width"] == 1280 and j["cluster_size"] == 2500


def test_gpus_world_size_mismatch_fails():
    r = _bench("--gpus", "2", "--launch-check", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_launcher_does_not_import_torch():
    # the launcher process must not be able to touch the GPU: torch is not even imported before the ranks are started
    code = ("import sys, runpy\n"
            "import moving_object_detector_amd.launch as L\n"
            "L.spawn_ranks = lambda *a, **k: (print('TORCH' if 'torch' in sys.modules else 'CLEAN'), 0)[1]\n"
            "sys.argv = ['bench.py', '--gpus', '4']\n"
            "runpy.run_path('bench.py', run_name='__main__')\n")
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)
    assert "CLEAN" in r.stdout, (r.stdout, r.stderr[-1500:])


def test_rank_command_is_the_drivers():
    from moving_object_detector_amd.launch import rank_command
    c = rank_command("bench.py", ["--gpus", "8"], 8, 29511)
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in c
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-3:] == ["bench.py", "--gpus", "8"]


def test_sequence_is_a_function_of_the_frame_index():
    from moving_object_detector_amd import synth
    _, full = synth.make_sequence(96, 64, 5, seed=4)
    assert full["disparity"].shape == (6, 64, 96) and full["flow"].shape == (5, 64, 96, 2)
    _, part = synth.make_sequence(96, 64, 2, seed=4, first=3)
    assert part["disparity"].tobytes() == full["disparity"][3:6].tobytes()
    assert part["flow"].tobytes() == full["flow"][3:5].tobytes()
    assert part["t"].tobytes() == full["t"][3:5].tobytes() and part["q"].tobytes() == full["q"][3:5].tobytes()
    d = full["disparity"]
    assert np.isnan(d).any() and (d == 0).any() and (d < 0).any() and (d > 128).any() and np.isnan(full["flow"]).any()


def test_shard_stream_chunks_plus_halo():
    from moving_object_detector_amd import dist as mdist
    from moving_object_detector_amd import synth
    _, s = synth.make_sequence(64, 48, 7, seed=2)
    mk = lambda first, frames: synth.make_sequence(64, 48, frames, seed=2, first=first)[1]
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            a = mdist.shard_stream(s["disparity"], s["flow"], s["t"], s["q"], s["dt"], r, world)
            b = mdist.local_stream(mk, 7, r, world)
            n = a["hi"] - a["lo"]
            assert (a["lo"], a["hi"]) == (b["lo"], b["hi"]) and len(a["disparity"]) == n + 1
            for k in ("disparity", "disparity_prev", "disparity_now", "flow", "t", "q", "dt"):
                assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), (world, r, k)
            # the halo: the first frame's previous plane is the plane before the chunk, i.e. the last "now" of the rank before
            if n:
                assert a["disparity_prev"][0].tobytes() == s["disparity"][a["lo"]].tobytes()
                assert a["disparity_now"][-1].tobytes() == s["disparity"][a["hi"]].tobytes()
            seen += list(range(a["lo"], a["hi"]))
        assert seen == list(range(7))


def test_launcher_without_gpus_flag_takes_world_size():
    # `torchrun --nproc-per-node 2 bench.py` with no --gpus in the script's arguments: WORLD_SIZE is the job's size
    from moving_object_detector_amd.launch import free_port
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--launch-check", "--distinct", "3"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["shards"] == [[0, 3], [3, 6]]


def test_one_rank_group_runs_the_same_collectives_on_gloo():
    # the group code of the N-rank run on a communicator of one (on the GPU box the same runs on RCCL: tests/test_gpu_rccl.py)
    code = ("import sys\n"
            "from moving_object_detector_amd import capi, synth, dist as mdist\n"
            "import torch, torch.distributed as dist\n"
            "assert mdist.broadcast_config('a', 'b') == ('a', 'b') and mdist.max_over_ranks(2.5) == 2.5   # no group: pass-through\n"
            "assert mdist.init_group('gloo', 0, 1) and mdist.group_is_up() and dist.get_world_size() == 1\n"
            "cam = capi.camera_struct(synth.make_camera(1280, 720)); prm = capi.params_struct(synth.Params(neighbor_distance=9))\n"
            "c2, p2 = mdist.broadcast_config(cam, prm)\n"
            "assert bytes(c2) == bytes(cam) and bytes(p2) == bytes(prm) and c2 is not cam\n"
            "assert mdist.max_over_ranks(1.25) == 1.25\n"
            "assert [t.tolist() for t in mdist.gather_counts(torch.tensor([3, 4], dtype=torch.int32))] == [[3, 4]]\n"
            "dist.destroy_process_group(); print('OK')\n")
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def _torchrun_bench(n, *argv, drop=("HSA_ENABLE_IPC_MODE_LEGACY",)):
    from moving_object_detector_amd.launch import free_port
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop):
        e.pop(k, None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), *argv],
                          env=e, capture_output=True, text=True, timeout=300)


def test_external_launcher_rank_gets_the_ipc_variable_before_torch_loads():
    # the driver's own command (torch.distributed.run ... bench.py) from an environment WITHOUT the variable: bench.py sets it at
    # import, before torch (and with it the ROCm runtime) is loaded, and takes the process group down before rank 0's solo legs
    r = _torchrun_bench(2, "--launch-check", "--distinct", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["ipc_env_set_before_torch"] is True and j["hsa_enable_ipc_mode_legacy"] == "0"
    assert j["group_down_before_solo_legs"] is True


def test_failed_rank_is_fatal_and_visible():
    r = _torchrun_bench(2, "--launch-check", "--distinct", "3", "--fail-rank", "1", drop=())
    assert r.returncode != 0
    assert "bench.py: rank 1 failed: RuntimeError" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no partial JSON


def test_cpu_baseline_uses_every_core_of_the_box():
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    assert "min(cores, 64)" not in src and "sched_getaffinity" in src
