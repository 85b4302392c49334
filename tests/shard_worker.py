"""Helper PROGRAM (not a test): one rank of a sharded stream on the real HIP path.

Started by tests/test_gpu_sharded_stream.py as `python tests/shard_worker.py rank world port out_dir W H total seed [defaults]`
(fresh child processes; the ranks rendezvous over gloo and may share cuda:0 on a one-GPU box).  Rank 0 owns the camera and the
parameters and broadcasts them; every rank materialises its chunk of the stream + the one-plane disparity halo
(dist.local_stream), runs mod_process_dev over it and saves planes / labels / objects for the parent to concatenate.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out_dir, W, H, total, seed = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    defaults = len(sys.argv) > 9 and sys.argv[9] == "defaults"       # the reference's default parameters (BASELINE config 4)
    import numpy as np
    import torch
    import torch.distributed as dist

    from moving_object_detector_amd import capi, synth
    from moving_object_detector_amd import dist as mdist
    from moving_object_detector_amd.pipeline import Context

    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        mdist.init_group("gloo", rank, world)       # both ranks share cuda:0 here; RCCL refuses two ranks on one device
    cam_s = prm_s = None
    if rank == 0:
        cam_s = capi.camera_struct(synth.make_camera(W, H))
        prm_s = capi.params_struct(synth.Params() if defaults else synth.Params(dynamic_flow_diff=2, cluster_size=120, neighbor_distance=4))
    cam_s, prm_s = mdist.broadcast_config(cam_s, prm_s, src=0)
    mk = lambda first, frames: synth.make_sequence(W, H, frames, seed=seed, first=first)[1]
    sh = mdist.local_stream(mk, total, rank, world)
    F = sh["hi"] - sh["lo"]
    dev = torch.device("cuda", 0)
    ctx = Context(W, H, max_frames=max(F, 1), device=0)
    ctx.set_camera(cam_s)
    ctx.set_params(prm_s)
    out = {"lo": sh["lo"], "hi": sh["hi"]}
    if F > 0:
        ws = ctx.workspace(F)
        D = torch.from_numpy(np.ascontiguousarray(sh["disparity"])).to(dev)      # chunk + halo, ONE buffer: prev = D, now = D + H*W
        fl = torch.from_numpy(np.ascontiguousarray(sh["flow"])).to(dev)
        b = ctx.make_batch(D[1:], D[:-1], fl, sh["t"], sh["q"], sh["dt"])
        assert ctx.process(b, ws) == 0
        ctx.synchronize()
        out["planes"] = ws["planes"].cpu().numpy()
        out["labels"] = ws["labels"].cpu().numpy()
        out["n_objects"] = ws["n_objects"].cpu().numpy()
        out["objects"] = ws["objects"].cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}of{world}.npz"), **out)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
