"""Helper PROGRAM (not a test): one fresh process that brings up a ONE-rank RCCL process group on cuda:0 and drives every
collective the N-rank run uses: the device broadcast of the {ModCamera, ModParams} block
(dist.broadcast_config), the device all-reduce (MAX) of the elapsed time (dist.max_over_ranks, what bench.py does after its
timed region), the all-gather of per-frame object counts (dist.gather_counts) and a barrier.  With one GPU per box this is
as much of RCCL as can execute: the communicator is created, librccl is loaded and its kernels run on the device.
Prints one JSON line.  usage: python tests/rccl_worker.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    import torch
    import torch.distributed as dist

    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd import dist as mdist

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert mdist.init_group("nccl", 0, 1, device=dev)
    cam = capi.camera_struct(synth.make_camera(1280, 720, "kitti"))
    prm = capi.params_struct(synth.Params(dynamic_flow_diff=3, cluster_size=777, neighbor_distance=7, depth_diff=0.21, dynamic_speed=0.4))
    cam2, prm2 = mdist.broadcast_config(cam, prm, src=0, device=dev)
    t = mdist.max_over_ranks(0.123456789, device=dev)
    counts = torch.arange(5, dtype=torch.int32, device=dev)
    g = mdist.gather_counts(counts)
    dist.barrier()
    with open("/proc/self/maps") as f:
        rccl = sorted({line.split()[-1] for line in f if "rccl" in line.lower()})
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "camera_roundtrip": bytes(cam2) == bytes(cam),
           "params_roundtrip": bytes(prm2) == bytes(prm), "max_over_ranks": t, "gathered": [x.cpu().tolist() for x in g],
           "rccl_libraries_mapped": rccl}
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
