"""ctypes face of oracle/libsgm_ref.so (oracle/sgm_ref.cpp).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class SgmParams(C.Structure):
    _fields_ = [("disparities", C.c_int32), ("p1", C.c_int32), ("p2", C.c_int32), ("paths", C.c_int32), ("lr_check", C.c_int32),
                ("median", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so, src = os.path.join(_HERE, "libsgm_ref.so"), os.path.join(_HERE, "sgm_ref.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libsgm_ref.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        u8, u32, u16, fp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_float)
        L.sgm_census.argtypes = [u8, C.c_int, C.c_int, u32]
        L.sgm_cost.argtypes = [u32, u32, C.c_int, C.c_int, C.c_int, u8]
        L.sgm_aggregate_path.argtypes = [u8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8]
        L.sgm_compute.argtypes = [u8, u8, C.c_int, C.c_int, C.POINTER(SgmParams), fp, u16]
        L.sgm_compute.restype = C.c_int
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def census(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.empty((H, W), np.uint32)
    lib().sgm_census(_p(img, C.c_uint8), W, H, _p(out, C.c_uint32))
    return out


def cost(cl, cr, D):
    cl, cr = np.ascontiguousarray(cl, np.uint32), np.ascontiguousarray(cr, np.uint32)
    H, W = cl.shape
    out = np.empty((H, W, D), np.uint8)
    lib().sgm_cost(_p(cl, C.c_uint32), _p(cr, C.c_uint32), W, H, D, _p(out, C.c_uint8))
    return out


def aggregate(Cv, P1, P2, direction):
    Cv = np.ascontiguousarray(Cv, np.uint8)
    H, W, D = Cv.shape
    out = np.empty((H, W, D), np.uint8)
    lib().sgm_aggregate_path(_p(Cv, C.c_uint8), W, H, D, P1, P2, direction, _p(out, C.c_uint8))
    return out


def compute(left, right, D=128, P1=6, P2=96, paths=8, lr_check=True, median=True, want_S=False):
    left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
    H, W = left.shape
    prm = SgmParams(D, P1, P2, paths, int(lr_check), int(median))
    disp = np.empty((H, W), np.float32)
    S = np.empty((H, W, D), np.uint16) if want_S else None
    rc = lib().sgm_compute(_p(left, C.c_uint8), _p(right, C.c_uint8), W, H, C.byref(prm), _p(disp, C.c_float),
                           _p(S, C.c_uint16) if want_S else None)
    assert rc == 0, rc
    return (disp, S) if want_S else disp
