"""ctypes face of oracle/liboracle.so (and oracle/_ref/liblookup_ref.so).

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("Tx", C.c_double), ("Ty", C.c_double), ("disp_f", C.c_float), ("disp_T", C.c_float),
                ("min_disparity", C.c_float), ("max_disparity", C.c_float)]


class OrcParams(C.Structure):
    _fields_ = [("dynamic_flow_diff", C.c_int32), ("cluster_size", C.c_int32), ("neighbor_distance", C.c_int32),
                ("reserved", C.c_int32), ("depth_diff", C.c_double), ("dynamic_speed", C.c_double)]


class OrcTransform(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4)]


class OrcObject(C.Structure):
    _fields_ = [("id", C.c_int32), ("n_points", C.c_int32), ("center", C.c_double * 3), ("orientation", C.c_double * 4),
                ("velocity", C.c_double * 3), ("bounding_box", C.c_double * 3)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "liblookup_ref.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        L.orc_get_point3d.argtypes = [C.POINTER(OrcCamera), fp, C.c_int, C.c_int, fp]
        L.orc_get_point3d.restype = C.c_int
        L.orc_rotation_from_transform.argtypes = [C.POINTER(OrcTransform), C.POINTER(C.c_double)]
        L.orc_norm3.argtypes = [C.c_float] * 3
        L.orc_norm3.restype = C.c_float
        L.orc_construct_faithful.argtypes = [C.POINTER(OrcCamera), C.POINTER(OrcParams), fp, fp, fp, C.POINTER(OrcTransform),
                                             C.c_double, C.c_void_p, fp, fp]
        L.orc_construct_faithful.restype = C.c_int
        L.orc_construct_tidy.argtypes = [C.POINTER(OrcCamera), C.POINTER(OrcParams), fp, fp, fp, C.POINTER(OrcTransform),
                                         C.c_double] + [fp] * 7
        L.orc_construct_tidy.restype = C.c_int
        L.orc_cluster_faithful.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(OrcParams), ip, C.POINTER(OrcObject),
                                           C.c_int, ip, ip]
        L.orc_cluster_faithful.restype = C.c_int
        L.orc_cluster_workspace_create.restype = C.c_void_p
        L.orc_cluster_workspace_destroy.argtypes = [C.c_void_p]
        L.orc_cluster_tidy.argtypes = [C.c_void_p] + [fp] * 6 + [C.c_int, C.c_int, C.POINTER(OrcParams), ip,
                                                                 C.POINTER(OrcObject), C.c_int, ip, ip]
        L.orc_cluster_tidy.restype = C.c_int
        L.orc_unpack_cloud.argtypes = [C.c_void_p, C.c_int64] + [fp] * 6
        L.orc_pack_cloud.argtypes = [C.c_void_p, C.c_int64] + [fp] * 6
        L.orc_lut_create.argtypes = [C.c_int64]
        L.orc_lut_create.restype = C.c_void_p
        L.orc_lut_destroy.argtypes = [C.c_void_p]
        L.orc_lut_reset.argtypes = [C.c_void_p]
        L.orc_lut_add_label.argtypes = [C.c_void_p]
        L.orc_lut_add_label.restype = C.c_int
        L.orc_lut_link.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_lut_lookup.argtypes = [C.c_void_p, C.c_int]
        L.orc_lut_lookup.restype = C.c_int
        L.orc_sorted_median_index.argtypes = [fp, C.c_int]
        L.orc_sorted_median_index.restype = C.c_int
        _lib = L
    return _lib


def ref_lookup_lib():
    """The reference's own LookupTable, or None when oracle/_ref was never built (no /root/reference, no prebuilt)."""
    build()
    p = os.path.join(_HERE, "_ref", "liblookup_ref.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.ref_lut_create.argtypes = [C.c_long]
    L.ref_lut_create.restype = C.c_void_p
    L.ref_lut_destroy.argtypes = [C.c_void_p]
    L.ref_lut_reset.argtypes = [C.c_void_p]
    L.ref_lut_add_label.argtypes = [C.c_void_p]
    L.ref_lut_add_label.restype = C.c_int
    L.ref_lut_link.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ref_lut_lookup.argtypes = [C.c_void_p, C.c_int]
    L.ref_lut_lookup.restype = C.c_int
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def camera_struct(cam) -> OrcCamera:
    return OrcCamera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.Tx, cam.Ty, float(cam.disp_f),
                     float(cam.disp_T), float(cam.min_disparity), float(cam.max_disparity))


def params_struct(prm) -> OrcParams:
    return OrcParams(int(prm.dynamic_flow_diff), int(prm.cluster_size), int(prm.neighbor_distance), 0,
                     float(prm.depth_diff), float(prm.dynamic_speed))


def transform_struct(t, q) -> OrcTransform:
    tf = OrcTransform()
    for i in range(3):
        tf.t[i] = float(t[i])
    for i in range(4):
        tf.q[i] = float(q[i])
    return tf


CLOUD_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("pad0", "f4"), ("vx", "f4"), ("vy", "f4"), ("vz", "f4"),
                        ("pad1", "f4")])


def depth_image(cam, d_now):
    """~depth: toDepthImage as construct() publishes it even when every other input is missing (the call below hands over
    only disparity_now and gets the "no flow" skip code back, scene_flow_constructor.cpp:110-123)."""
    L = lib()
    H, W = d_now.shape
    c, p = camera_struct(cam), params_struct(type("P", (), dict(dynamic_flow_diff=5, cluster_size=2500, neighbor_distance=4,
                                                                 depth_diff=0.15, dynamic_speed=0.3))())
    d_now = np.ascontiguousarray(d_now, np.float32)
    depth = np.empty((H, W), np.float32)
    rc = L.orc_construct_faithful(C.byref(c), C.byref(p), _fp(d_now), None, None, None, 0.0, None, None, _fp(depth))
    assert rc == 3, rc
    return depth


def construct(cam, prm, d_now, d_prev, flow, t, q, dt, mode="faithful"):
    """Returns dict of float32 (H,W) planes x,y,z,vx,vy,vz, the 32-byte AoS cloud, the static flow and the depth image."""
    L = lib()
    H, W = d_now.shape
    c, p, tf = camera_struct(cam), params_struct(prm), transform_struct(t, q)
    d_now = np.ascontiguousarray(d_now, np.float32)
    d_prev = np.ascontiguousarray(d_prev, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    sflow = np.empty((H, W, 2), np.float32)
    planes = {k: np.empty((H, W), np.float32) for k in ("x", "y", "z", "vx", "vy", "vz")}
    if mode == "faithful":
        cloud = np.zeros((H, W), CLOUD_DTYPE)
        depth = np.empty((H, W), np.float32)
        rc = L.orc_construct_faithful(C.byref(c), C.byref(p), _fp(d_now), _fp(d_prev), _fp(flow), C.byref(tf), dt,
                                      cloud.ctypes.data, _fp(sflow), _fp(depth))
        assert rc == 0, rc
        for k in planes:
            planes[k] = np.ascontiguousarray(cloud[k])
        planes["depth"] = depth
    else:
        rc = L.orc_construct_tidy(C.byref(c), C.byref(p), _fp(d_now), _fp(d_prev), _fp(flow), C.byref(tf), dt,
                                  *[_fp(planes[k]) for k in ("x", "y", "z", "vx", "vy", "vz")], _fp(sflow))
        assert rc == 0, rc
        cloud = pack_cloud(planes)
        planes["depth"] = depth_image(cam, d_now)
    planes["cloud"] = cloud
    planes["static_flow"] = sflow
    return planes


def pack_cloud(planes) -> np.ndarray:
    H, W = planes["x"].shape
    cloud = np.zeros((H, W), CLOUD_DTYPE)
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        cloud[k] = planes[k]
    return cloud


def _objects_to_list(objs, n, amb):
    out = []
    for i in range(n):
        o = objs[i]
        out.append({"id": o.id, "n_points": o.n_points, "center": np.array(o.center[:]), "velocity": np.array(o.velocity[:]),
                    "bounding_box": np.array(o.bounding_box[:]), "orientation": np.array(o.orientation[:]),
                    "ambiguous": bool(amb[i])})
    return out


def cluster(cloud_or_planes, prm, mode="faithful", max_objects=4096, workspace=None):
    """Returns (labels (H,W) int32, objects list, K)."""
    L = lib()
    p = params_struct(prm)
    objs = (OrcObject * max_objects)()
    nobj = C.c_int32(0)
    amb = np.zeros(max_objects, np.int32)
    ip = C.POINTER(C.c_int32)
    if mode == "faithful":
        cloud = cloud_or_planes if isinstance(cloud_or_planes, np.ndarray) else pack_cloud(cloud_or_planes)
        cloud = np.ascontiguousarray(cloud)
        H, W = cloud.shape
        labels = np.empty((H, W), np.int32)
        K = L.orc_cluster_faithful(cloud.ctypes.data, W, H, C.byref(p), labels.ctypes.data_as(ip), objs, max_objects,
                                   C.byref(nobj), amb.ctypes.data_as(ip))
    else:
        pl = cloud_or_planes
        H, W = pl["x"].shape
        labels = np.empty((H, W), np.int32)
        own = workspace is None
        ws = L.orc_cluster_workspace_create() if own else workspace
        arrs = [np.ascontiguousarray(pl[k], np.float32) for k in ("x", "y", "z", "vx", "vy", "vz")]
        K = L.orc_cluster_tidy(ws, *[_fp(a) for a in arrs], W, H, C.byref(p), labels.ctypes.data_as(ip), objs, max_objects,
                               C.byref(nobj), amb.ctypes.data_as(ip))
        if own:
            L.orc_cluster_workspace_destroy(ws)
    return labels, _objects_to_list(objs, min(nobj.value, max_objects), amb), K
