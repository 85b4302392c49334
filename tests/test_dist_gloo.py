"""CPU, world_size 2 over gloo: the N>1 path = intrinsics/params broadcast from rank 0 + frame sharding with no
data-path collective.  The per-rank compute is stood in for by the CPU oracle here (there is no GPU
in this container); on GPUs the same code drives libmod_sf.so with backend "nccl" (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd import dist as mdist
    from oracle import pyoracle
    W, H = 96, 64
    cam_s = prm_s = None
    if rank == 0:   # only rank 0 knows the camera and the reconfigured parameters
        cam_s = capi.camera_struct(synth.make_camera(W, H))
        prm_s = capi.params_struct(synth.Params(dynamic_flow_diff=1, cluster_size=17, neighbor_distance=3))
    cam_s, prm_s = mdist.broadcast_config(cam_s, prm_s, src=0)
    assert cam_s.width == W and prm_s.cluster_size == 17 and prm_s.neighbor_distance == 3
    lo, hi = mdist.shard_range(total_frames, rank, world)
    prm = synth.Params(prm_s.dynamic_flow_diff, prm_s.cluster_size, prm_s.neighbor_distance, prm_s.depth_diff, prm_s.dynamic_speed)
    cam = synth.make_camera(cam_s.width, cam_s.height)
    counts = []
    for fidx in range(lo, hi):
        _, f = synth.make_frame(W, H, seed=9, frame=fidx)
        sf = pyoracle.construct(cam, prm, f.disparity_now, f.disparity_prev, f.flow, f.translation, f.quaternion, f.dt, "tidy")
        labels, objs, K = pyoracle.cluster(sf, prm, "tidy")
        counts.append(len(objs))
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), np.array([lo, hi] + counts))
    # optional ordered view on rank 0
    g = mdist.gather_counts(torch.tensor(counts[: total_frames // world], dtype=torch.int32))
    assert len(g) == world
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_broadcast_and_shard(tmp_path):
    world, total = 2, 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    seen = []
    for r in range(world):
        a = np.load(tmp_path / f"counts_{r}.npy")
        seen += list(range(int(a[0]), int(a[1])))
        assert len(a) - 2 == int(a[1]) - int(a[0])
    assert seen == list(range(total))          # shards tile the stream exactly once, in order


def test_shard_range_properties():
    from moving_object_detector_amd.dist import shard_range
    for total in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_config_block_roundtrip():
    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd.dist import pack_config, unpack_config
    cam = capi.camera_struct(synth.make_camera(1280, 720))
    prm = capi.params_struct(synth.Params(neighbor_distance=7))
    c2, p2 = unpack_config(pack_config(cam, prm))
    assert bytes(c2) == bytes(cam) and bytes(p2) == bytes(prm)
