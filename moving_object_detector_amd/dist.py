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


def broadcast_config(cam, prm, src: int = 0, device=None):
    """Rank `src` supplies (cam, prm); every rank returns the same structs.  Other ranks may pass None."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


def gather_counts(local_counts: torch.Tensor):
    """Optional ordered view for rank 0: all-gather of per-frame object counts (equal shard sizes)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_counts]
    out = [torch.empty_like(local_counts) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local_counts)
    return out
