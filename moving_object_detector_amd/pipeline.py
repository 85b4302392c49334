"""Device-resident driver of the C ABI: owns a ModContext and the HBM buffers (torch tensors are used purely as
device memory + stream plumbing; every computation happens in libmod_sf.so)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import capi

PLANES = ("x", "y", "z", "vx", "vy", "vz")

OBJECT_DTYPE = np.dtype([("id", "<i4"), ("n_points", "<i4"), ("center", "<f8", 3), ("orientation", "<f8", 4),
                         ("velocity", "<f8", 3), ("bounding_box", "<f8", 3)])
assert OBJECT_DTYPE.itemsize == capi.MOD_OBJECT_BYTES


# Planes of one call are carved from ONE block with their bases 1 MiB further apart than their size: at 512 x 1280 x 720 a plane is a
# multiple of 8 MiB, so back to back the nine streams of a scene-flow wave (3 in, 6 out) would carry the same low address bits — the
# same DRAM bank group at the same moment.  Measured inside one process (tools/ab_skew.py): -1 ... -3 % on the scene-flow kernel for
# 128 KiB ... 1 MiB of stagger (multiples of 2 MiB: nothing), depending on how (un)lucky the box's back-to-back placement is.
PLANE_STAGGER_BYTES = 1 << 20


def staggered(sizes, dtype, device, stagger_bytes: int = PLANE_STAGGER_BYTES):
    """1-D tensors of `sizes` elements each, carved from one allocation, consecutive bases `stagger_bytes` apart beyond their sizes."""
    skew = stagger_bytes // torch.empty((), dtype=dtype).element_size()
    flat = torch.empty(sum(sizes) + skew * (len(sizes) - 1), dtype=dtype, device=device)
    out, at = [], 0
    for n in sizes:
        out.append(flat[at:at + n])
        at += n + skew
    return out


class Context:
    """One context per GPU / stream (single caller), like one SceneFlowConstructor + one ClustererNodelet."""

    def __init__(self, width: int, height: int, max_frames: int = 1, device: int = 0, max_objects: int = 0,
                 use_torch_stream: bool = True, batch_chunks: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the MI355X path has no CPU fallback")
        self.lib = capi.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.width, self.height, self.max_frames = width, height, max_frames
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else None
        cfg = capi.ModConfig(device, width, height, max_frames, max_objects, batch_chunks, stream)
        h = C.c_void_p()
        rc = self.lib.mod_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise capi.ModError(rc, "mod_create failed")
        self.h = h
        self.max_objects = max_objects if max_objects > 0 else max(1, (width * height) // 100)
        self.mask_words = (width + 63) // 64
        self._ws = None

    # ---- configuration ----------------------------------------------------------------------------------------
    def _check(self, rc: int) -> int:
        if rc < 0:
            raise capi.ModError(rc, self.lib.mod_last_error(self.h).decode())
        return rc

    def set_camera(self, cam) -> None:
        s = cam if isinstance(cam, capi.ModCamera) else capi.camera_struct(cam)
        self._check(self.lib.mod_set_camera(self.h, C.byref(s)))
        self.width, self.height = s.width, s.height
        self.mask_words = (s.width + 63) // 64

    def set_params(self, prm) -> None:
        s = prm if isinstance(prm, capi.ModParams) else capi.params_struct(prm)
        self._check(self.lib.mod_set_params(self.h, C.byref(s)))

    def get_camera(self) -> capi.ModCamera:
        s = capi.ModCamera()
        self._check(self.lib.mod_get_camera(self.h, C.byref(s)))
        return s

    def get_params(self) -> capi.ModParams:
        s = capi.ModParams()
        self._check(self.lib.mod_get_params(self.h, C.byref(s)))
        return s

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.mod_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers ----------------------------------------------------------------------------------------------
    def workspace(self, frames: int, aos: bool = False, extras: bool = False, labels: bool = True, xy: bool = True) -> dict:
        """Output buffers for `frames` frames, allocated once and reused.  labels=False: no cluster-label plane (the reference
        renders its cluster image only for subscribers, clusterer_nodelet.cpp:235-236).  xy=False: process() hands the library no x / y
        planes (nobody takes the cloud, scene_flow_constructor.cpp:141-142): the rows 0 and 1 of `planes` are then never written."""
        key = (frames, self.height, self.width, aos, extras, labels, xy)
        if self._ws is not None and self._ws["key"] == key:
            return self._ws
        H, W, dev = self.height, self.width, self.device
        ws = {"key": key, "xy": xy}
        per = frames * H * W                             # six planes from one block, bases staggered (see PLANE_STAGGER_BYTES)
        skew = PLANE_STAGGER_BYTES // 4
        ws["planes"] = torch.empty(6 * per + 5 * skew, dtype=torch.float32, device=dev).as_strided((6, frames, H, W), (per + skew, H * W, W, 1))
        ws["mask"] = torch.empty((frames, H, self.mask_words), dtype=torch.int64, device=dev)
        ws["labels"] = torch.empty((frames, H, W), dtype=torch.int32, device=dev) if labels else None
        ws["objects"] = torch.zeros((frames, self.max_objects, capi.MOD_OBJECT_BYTES), dtype=torch.uint8, device=dev)
        ws["n_objects"] = torch.zeros((frames,), dtype=torch.int32, device=dev)
        ws["n_clusters"] = torch.zeros((frames,), dtype=torch.int32, device=dev)
        ws["aos"] = torch.empty((frames, H, W, 8), dtype=torch.float32, device=dev) if aos else None
        ws["depth"] = torch.empty((frames, H, W), dtype=torch.float32, device=dev) if extras else None
        ws["static_flow"] = torch.empty((frames, H, W, 2), dtype=torch.float32, device=dev) if extras else None
        self._ws = ws
        return ws

    def _planes_struct(self, ws, with_mask=True) -> capi.ModSceneFlowPlanes:
        p = ws["planes"]
        s = capi.ModSceneFlowPlanes()
        for i, k in enumerate(PLANES):
            setattr(s, k, p[i].data_ptr() if (ws.get("xy", True) or k not in ("x", "y")) else None)
        s.dynamic_mask = ws["mask"].data_ptr() if with_mask else None
        s.cloud_aos = ws["aos"].data_ptr() if ws.get("aos") is not None else None
        s.depth = ws["depth"].data_ptr() if ws.get("depth") is not None else None
        s.static_flow = ws["static_flow"].data_ptr() if ws.get("static_flow") is not None else None
        return s

    @staticmethod
    def _cluster_struct(ws) -> capi.ModClusterOut:
        return capi.ModClusterOut(ws["labels"].data_ptr() if ws["labels"] is not None else None, ws["objects"].data_ptr(), ws["n_objects"].data_ptr(),
                                  ws["n_clusters"].data_ptr())

    def make_batch(self, d_now: torch.Tensor, d_prev: Optional[torch.Tensor], flow: Optional[torch.Tensor], ts, qs, dts):
        """ModFrameBatch over device tensors (F,H,W), (F,H,W), (F,H,W,2) + host transforms.  Keeps references alive."""
        F = d_now.shape[0]
        b = capi.ModFrameBatch()
        b.frames = F
        b.disparity_now = d_now.data_ptr()
        b.disparity_prev = d_prev.data_ptr() if d_prev is not None else None
        b.flow = flow.data_ptr() if flow is not None else None
        tr = capi.transforms_array(ts, qs) if ts is not None else None
        dt = (C.c_double * F)(*[float(v) for v in dts]) if dts is not None else None
        b.transforms = tr
        b.dt = dt
        b._keep = (d_now, d_prev, flow, tr, dt)
        return b

    # ---- hot path ---------------------------------------------------------------------------------------------
    def scene_flow(self, batch, ws) -> int:
        return self._check(self.lib.mod_scene_flow_dev(self.h, C.byref(batch), C.byref(self._planes_struct(ws))))

    def cluster(self, frames: int, ws, mask_ready: bool) -> int:
        pl = self._planes_struct(ws, with_mask=mask_ready)
        return self._check(self.lib.mod_cluster_dev(self.h, frames, C.byref(pl), C.byref(self._cluster_struct(ws))))

    def process(self, batch, ws) -> int:
        return self._check(self.lib.mod_process_dev(self.h, C.byref(batch), C.byref(self._planes_struct(ws)),
                                                    C.byref(self._cluster_struct(ws))))

    def synchronize(self) -> None:
        self._check(self.lib.mod_synchronize(self.h))

    # ---- measurement ------------------------------------------------------------------------------------------
    def set_profiling(self, on, stages=None) -> None:
        """Stage timers on/off; `stages` (iterable of MOD_STAGE_*) restricts them to those stages."""
        mask = 0
        if on:
            mask = capi.MOD_PROFILE_ALL if stages is None else sum(1 << int(s) for s in set(stages))
        self._check(self.lib.mod_set_profiling(self.h, mask))

    def reset_stage_times(self) -> None:
        self._check(self.lib.mod_reset_stage_times(self.h))

    def stage_time(self, stage: int):
        ms, calls = C.c_double(0), C.c_int64(0)
        self._check(self.lib.mod_get_stage_time(self.h, stage, C.byref(ms), C.byref(calls)))
        return ms.value, calls.value

    # ---- results to host --------------------------------------------------------------------------------------
    @staticmethod
    def objects_to_host(ws) -> list:
        """Per frame: structured numpy array of the accepted ModObject records."""
        n = ws["n_objects"].cpu().numpy()
        raw = ws["objects"].cpu().numpy()
        out = []
        for f in range(raw.shape[0]):
            k = min(int(n[f]), raw.shape[1])
            out.append(np.frombuffer(raw[f, :k].tobytes(), dtype=OBJECT_DTYPE).copy())
        return out
