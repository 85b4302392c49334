"""ctypes binding of libmod_sf.so (include/mod_sf.h).

The library is the product; there is no CPU fallback.  Loading fails loudly when the shared object has not been
built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C moving_object_detector_amd/csrc``).

torch is imported *before* the library on purpose: PyTorch-ROCm bundles its own ``libamdhip64.so`` (same soname as the
system one), and the dynamic loader then resolves our library's HIP dependency to that already-loaded runtime, so device
pointers and streams owned by torch are valid inside our kernels (one HIP runtime per process).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOD_SF_LIB") or os.path.join(_HERE, "libmod_sf.so")   # MOD_SF_LIB: A/B builds during development

MOD_OK = 0
MOD_SKIP_NO_DISPARITY_NOW = 1
MOD_SKIP_NO_DISPARITY_PREV = 2
MOD_SKIP_NO_FLOW = 3
MOD_SKIP_NO_TRANSFORM = 4
MOD_ERR_INVALID_ARGUMENT = -1
MOD_ERR_NOT_CONFIGURED = -2
MOD_ERR_CAPACITY = -3
MOD_ERR_DEVICE = -4
MOD_ERR_NO_DEVICE = -5
MOD_STAGE_SCENE_FLOW, MOD_STAGE_CCL_TILE, MOD_STAGE_CCL_LINK, MOD_STAGE_CCL_MERGE = 0, 1, 2, 3
MOD_STAGE_FINAL, MOD_STAGE_MEDIAN, MOD_STAGE_CLUSTER_GROUP, MOD_STAGE_COUNT = 4, 5, 6, 7
MOD_PROFILE_ALL = 0x7F
MOD_PER_KERNEL_CLUSTER_STAGES = (1, 2, 3, 4, 5)
MOD_PIPELINE_DEPTH = 3
STAGE_NAMES = ("k_scene_flow", "k_ccl_bits+k_ccl_tile_list", "k_ccl_link", "k_ccl_merge", "k_final", "k_median+k_median_ties",
               "cluster group (first launch to last)")

# every symbol include/mod_sf.h declares (tests check that the library exports all of them)
EXPORTS = [
    "mod_abi_version", "mod_create", "mod_destroy", "mod_last_error", "mod_set_camera", "mod_set_params",
    "mod_get_camera", "mod_get_params", "mod_synchronize", "mod_scene_flow_dev", "mod_dynamic_mask_dev",
    "mod_cluster_dev", "mod_process_dev", "mod_pack_cloud_dev", "mod_unpack_cloud_dev", "mod_process_frame_host",
    "mod_cluster_cloud_host", "mod_submit_frame_host", "mod_submit_stereo_host", "mod_collect_frame_host", "mod_forget_previous", "mod_host_malloc", "mod_host_free",
    "mod_malloc", "mod_free", "mod_memcpy_h2d", "mod_memcpy_d2h", "mod_set_profiling",
    "mod_get_stage_time", "mod_reset_stage_times", "mod_depth_image_dev", "mod_depth_image_host", "mod_static_flow_host",
    "mod_sgm_census_dev", "mod_sgm_path_dev", "mod_sgm_compute_dev", "mod_sgm_compute_host",
]


class ModConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32), ("max_frames", C.c_int32),
                ("max_objects", C.c_int32), ("batch_chunks", C.c_int32), ("stream", C.c_void_p)]


class ModCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("Tx", C.c_double), ("Ty", C.c_double), ("disp_f", C.c_float), ("disp_T", C.c_float),
                ("min_disparity", C.c_float), ("max_disparity", C.c_float)]


class ModParams(C.Structure):
    _fields_ = [("dynamic_flow_diff", C.c_int32), ("cluster_size", C.c_int32), ("neighbor_distance", C.c_int32),
                ("reserved", C.c_int32), ("depth_diff", C.c_double), ("dynamic_speed", C.c_double)]


class ModTransform(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4)]


class ModObject(C.Structure):
    _fields_ = [("id", C.c_int32), ("n_points", C.c_int32), ("center", C.c_double * 3), ("orientation", C.c_double * 4),
                ("velocity", C.c_double * 3), ("bounding_box", C.c_double * 3)]


class ModFrameBatch(C.Structure):
    _fields_ = [("frames", C.c_int32), ("reserved", C.c_int32), ("disparity_now", C.c_void_p), ("disparity_prev", C.c_void_p),
                ("flow", C.c_void_p), ("transforms", C.POINTER(ModTransform)), ("dt", C.POINTER(C.c_double))]


class ModSceneFlowPlanes(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("vx", C.c_void_p), ("vy", C.c_void_p),
                ("vz", C.c_void_p), ("dynamic_mask", C.c_void_p), ("cloud_aos", C.c_void_p), ("depth", C.c_void_p),
                ("static_flow", C.c_void_p)]


class ModSgmParams(C.Structure):
    _fields_ = [("disparities", C.c_int32), ("p1", C.c_int32), ("p2", C.c_int32), ("paths", C.c_int32), ("lr_check", C.c_int32),
                ("median", C.c_int32)]


class ModClusterOut(C.Structure):
    _fields_ = [("labels", C.c_void_p), ("objects", C.c_void_p), ("n_objects", C.c_void_p), ("n_clusters", C.c_void_p)]


MOD_OBJECT_BYTES = C.sizeof(ModObject)
assert MOD_OBJECT_BYTES == 112


class ModError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmod_sf error {code}: {msg}")
        self.code = code


_lib = None


def load(require_torch_first: bool = True):
    """Load libmod_sf.so.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                          f"(run __graft_entry__.build() or `make -C moving_object_detector_amd/csrc`). "
                          f"There is no CPU fallback.")
    if require_torch_first:
        import torch  # noqa: F401  (see module docstring: one HIP runtime per process)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p, dp = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)
    L.mod_abi_version.restype = C.c_int
    L.mod_create.argtypes = [C.POINTER(ModConfig), C.POINTER(vp)]
    L.mod_destroy.argtypes = [vp]
    L.mod_destroy.restype = None
    L.mod_last_error.argtypes = [vp]
    L.mod_last_error.restype = C.c_char_p
    L.mod_set_camera.argtypes = [vp, C.POINTER(ModCamera)]
    L.mod_set_params.argtypes = [vp, C.POINTER(ModParams)]
    L.mod_get_camera.argtypes = [vp, C.POINTER(ModCamera)]
    L.mod_get_params.argtypes = [vp, C.POINTER(ModParams)]
    L.mod_synchronize.argtypes = [vp]
    L.mod_scene_flow_dev.argtypes = [vp, C.POINTER(ModFrameBatch), C.POINTER(ModSceneFlowPlanes)]
    L.mod_depth_image_dev.argtypes = [vp, i32, vp, vp]
    L.mod_depth_image_host.argtypes = [vp, vp, vp]
    L.mod_static_flow_host.argtypes = [vp, vp, C.POINTER(ModTransform), vp]
    L.mod_sgm_census_dev.argtypes = [vp, i32, vp, vp]
    L.mod_sgm_compute_dev.argtypes = [vp, i32, vp, vp, C.POINTER(ModSgmParams), vp]
    L.mod_sgm_compute_host.argtypes = [vp, vp, vp, C.POINTER(ModSgmParams), vp]
    L.mod_sgm_path_dev.argtypes = [vp, i32, vp, vp, C.POINTER(ModSgmParams), i32, vp, vp]
    L.mod_dynamic_mask_dev.argtypes = [vp, i32, vp, vp, vp, vp]
    L.mod_cluster_dev.argtypes = [vp, i32, C.POINTER(ModSceneFlowPlanes), C.POINTER(ModClusterOut)]
    L.mod_process_dev.argtypes = [vp, C.POINTER(ModFrameBatch), C.POINTER(ModSceneFlowPlanes), C.POINTER(ModClusterOut)]
    L.mod_pack_cloud_dev.argtypes = [vp, i32, C.POINTER(ModSceneFlowPlanes), vp]
    L.mod_unpack_cloud_dev.argtypes = [vp, i32, vp, C.POINTER(ModSceneFlowPlanes)]
    L.mod_process_frame_host.argtypes = [vp, vp, vp, vp, C.POINTER(ModTransform), C.c_double, vp, vp, vp, i32,
                                         C.POINTER(i32)]
    L.mod_cluster_cloud_host.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
    L.mod_submit_frame_host.argtypes = [vp, vp, vp, vp, C.POINTER(ModTransform), C.c_double, vp, vp, vp, i32, C.POINTER(i32)]
    L.mod_submit_stereo_host.argtypes = [vp, vp, vp, C.POINTER(ModSgmParams), vp, C.POINTER(ModTransform), C.c_double, vp, vp, vp, i32, vp,
                                         C.POINTER(i32)]
    L.mod_collect_frame_host.argtypes = [vp, i32, C.POINTER(i32)]
    L.mod_forget_previous.argtypes = [vp]
    L.mod_host_malloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.mod_host_free.argtypes = [vp, vp]
    L.mod_malloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.mod_free.argtypes = [vp, vp]
    L.mod_memcpy_h2d.argtypes = [vp, vp, vp, C.c_uint64]
    L.mod_memcpy_d2h.argtypes = [vp, vp, vp, C.c_uint64]
    L.mod_set_profiling.argtypes = [vp, i32]
    L.mod_get_stage_time.argtypes = [vp, i32, dp, i64p]
    L.mod_reset_stage_times.argtypes = [vp]
    for name in EXPORTS:
        if name not in ("mod_destroy", "mod_last_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def camera_struct(cam) -> ModCamera:
    return ModCamera(int(cam.width), int(cam.height), float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy),
                     float(cam.Tx), float(cam.Ty), float(cam.disp_f), float(cam.disp_T), float(cam.min_disparity),
                     float(cam.max_disparity))


def params_struct(prm) -> ModParams:
    return ModParams(int(prm.dynamic_flow_diff), int(prm.cluster_size), int(prm.neighbor_distance), 0,
                     float(prm.depth_diff), float(prm.dynamic_speed))


def transforms_array(ts, qs):
    n = len(ts)
    arr = (ModTransform * n)()
    for i in range(n):
        for k in range(3):
            arr[i].t[k] = float(ts[i][k])
        for k in range(4):
            arr[i].q[k] = float(qs[i][k])
    return arr
