"""Multi-GPU plumbing: frames are the unit of sharding.

Frame t only needs {disparity_t, disparity_{t-1}, flow_t, T_t, dt_t}, so every rank runs the whole pipeline on its
own frames and there is no data-path collective.  The one exchange the path has is the camera-intrinsics + parameter
block (~130 B) that rank 0 owns (in the reference CameraInfo arrives with every frame and reconfigure requests hit one
node, scene_flow_constructor.cpp:368-375,401-407): it is broadcast once per stream / reconfigure over RCCL
(``torch.distributed`` backend "nccl" on ROCm) — or gloo in the CPU tests.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi

_BLOCK_BYTES = C.sizeof(capi.ModCamera) + C.sizeof(capi.ModParams)


def pack_config(cam: capi.ModCamera, prm: capi.ModParams) -> np.ndarray:
    return np.frombuffer(bytes(cam) + bytes(prm), dtype=np.uint8).copy()


def unpack_config(block: np.ndarray):
    raw = block.tobytes()
    cam = capi.ModCamera.from_buffer_copy(raw[:C.sizeof(capi.ModCamera)])
    prm = capi.ModParams.from_buffer_copy(raw[C.sizeof(capi.ModCamera):_BLOCK_BYTES])
    return cam, prm


def init_group(backend: str, rank: int, world: int, device=None, port: int | None = None) -> bool:
    """Joins the job's process group — also when the job has ONE rank, so that the single-GPU run drives the very code the
    N-rank run does (RCCL communicator, device broadcast, device all-reduce).  `backend` "nccl" is RCCL on ROCm and wants
    `device` (its `device_id`); "gloo" is for CPU tests / rehearsals.  MASTER_ADDR / MASTER_PORT come from the launcher; a
    lone rank without one rendezvouses with itself on 127.0.0.1.  Returns True when a group is up."""
    if not dist.is_available():
        return False
    if dist.is_initialized():
        return True
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world != 1 and port is None:
            raise RuntimeError("MASTER_PORT is not set: start the ranks with a launcher (torch.distributed.run, launch.spawn_ranks)")
        from .launch import free_port
        os.environ["MASTER_PORT"] = str(port or free_port())
    # (HSA_ENABLE_IPC_MODE_LEGACY=0, which RCCL needs across processes on this pool's driver, has to be in the environment BEFORE
    # the ROCm runtime loads, i.e. before `import torch`: bench.py sets it at import, launch.spawn_ranks in the children's
    # environment; setting it here would be too late to matter.)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return True


def group_is_up() -> bool:
    return dist.is_available() and dist.is_initialized()


def max_over_ranks(value: float, device=None) -> float:
    """The job's time is the slowest rank's: all-reduce (MAX) of one F64, on `device` when the backend is RCCL."""
    if not group_is_up():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_config(cam, prm, src: int = 0, device=None):
    """Rank `src` supplies (cam, prm); every rank returns the same structs.  Other ranks may pass None.  Runs whenever a
    process group is up — a one-rank group included, where the broadcast still goes through the backend's communicator."""
    if not group_is_up():
        return cam, prm
    t = torch.zeros(_BLOCK_BYTES, dtype=torch.uint8, device=device if device is not None else "cpu")
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(pack_config(cam, prm)))
    dist.broadcast(t, src=src)
    return unpack_config(t.cpu().numpy())


def shard_range(total_frames: int, rank: int, world: int):
    """Contiguous chunk [lo, hi) of a stream for `rank`; chunks differ by at most one frame."""
    base, rem = divmod(total_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_stream(disparity, flow, t, q, dt, rank: int, world: int):
    """Rank `rank`'s share of a stream: a contiguous chunk of frames plus ONE disparity plane of halo.

    A stream of F frames is `disparity` [F+1] planes (frame i pairs previous = disparity[i] with now = disparity[i+1], the
    reference's `disparity_previous_ = disparity_now_`, scene_flow_constructor.cpp:397-398) and per-frame `flow`, `t`, `q`, `dt`
    [F].  The chunk [lo, hi) of shard_range needs planes lo .. hi: its first frame's "previous" plane is the halo — the plane
    the rank before it produced last.  Works on anything sliceable along axis 0 (numpy arrays, torch tensors); returns views:
    {"lo", "hi", "disparity" [hi-lo+1], "disparity_prev" [hi-lo] (= disparity[:-1]), "disparity_now" [hi-lo] (= disparity[1:]),
     "flow", "t", "q", "dt" [hi-lo]}.
    """
    F = len(flow)
    if len(disparity) != F + 1 or len(t) != F or len(q) != F or len(dt) != F:
        raise ValueError("a stream of F frames has F+1 disparity planes and F flows / transforms / dts")
    lo, hi = shard_range(F, rank, world)
    d = disparity[lo:hi + 1]
    return {"lo": lo, "hi": hi, "disparity": d, "disparity_prev": d[:hi - lo], "disparity_now": d[1:hi - lo + 1],
            "flow": flow[lo:hi], "t": t[lo:hi], "q": q[lo:hi], "dt": dt[lo:hi]}


def local_stream(make, total_frames: int, rank: int, world: int):
    """shard_stream for a stream that is a deterministic function of its frame index: `make(first, frames)` returns
    {"disparity" [frames+1], "flow", "t", "q", "dt" [frames]} for stream frames first .. first+frames-1 (e.g.
    synth.make_sequence), so each rank materialises only its chunk + the one-plane halo.  Same dict as shard_stream."""
    lo, hi = shard_range(total_frames, rank, world)
    s = make(lo, hi - lo)
    return shard_stream(s["disparity"], s["flow"], s["t"], s["q"], s["dt"], 0, 1) | {"lo": lo, "hi": hi}


def gather_counts(local_counts: torch.Tensor):
    """Optional ordered view for rank 0: all-gather of per-frame object counts (equal shard sizes)."""
    if not group_is_up():
        return [local_counts]
    out = [torch.empty_like(local_counts) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local_counts)
    return out
