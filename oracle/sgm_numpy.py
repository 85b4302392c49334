"""Second, independently written restatement of the SGM disparity estimator (see oracle/sgm_ref.cpp for the algorithm, its
source and the "parity unpinned" note).  TEST INFRASTRUCTURE ONLY.  Vectorised numpy: the two are checked against each other
(tests/test_sgm_oracle.py) and this one generates the committed fixtures (tests/golden/make_sgm_golden.py).
"""
from __future__ import annotations

import numpy as np

_POP16 = np.array([bin(i).count("1") for i in range(1 << 16)], np.uint8)


def popcount32(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.uint32)
    return (_POP16[a & 0xFFFF] + _POP16[a >> 16]).astype(np.uint8)


def census(img: np.ndarray) -> np.ndarray:
    """Centre-symmetric 9 x 7 census, 31 bits, border pixels 0."""
    H, W = img.shape
    out = np.zeros((H, W), np.uint32)
    if H < 7 or W < 9:
        return out
    I = img.astype(np.int32)
    acc = np.zeros((H - 6, W - 8), np.uint32)
    pairs = [(dy, dx) for dy in (-3, -2, -1) for dx in range(-4, 5)] + [(0, dx) for dx in range(-4, 0)]
    for dy, dx in pairs:
        a = I[3 + dy:H - 3 + dy, 4 + dx:W - 4 + dx]
        b = I[3 - dy:H - 3 - dy, 4 - dx:W - 4 - dx]
        acc = (acc << np.uint32(1)) | (a >= b).astype(np.uint32)
    out[3:H - 3, 4:W - 4] = acc
    return out


def cost_volume(cl: np.ndarray, cr: np.ndarray, D: int) -> np.ndarray:
    H, W = cl.shape
    C = np.full((H, W, D), 31, np.uint8)
    for d in range(D):
        if d < W:
            C[:, d:, d] = popcount32(cl[:, d:] ^ cr[:, :W - d])
    return C


_DIRS = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (-1, 1), (1, -1)]


def aggregate(C: np.ndarray, P1: int, P2: int, direction: int) -> np.ndarray:
    """One path, column by column (or row by row for the vertical ones); int16 arithmetic, result uint8."""
    H, W, D = C.shape
    rx, ry = _DIRS[direction]
    Ci = C.astype(np.int16)
    L = np.zeros((H, W, D), np.int16)

    def step(cur, prev, valid):
        """cur, prev: (n, D); valid: (n,) bool — rows whose predecessor exists."""
        mn = prev.min(axis=1, keepdims=True)
        best = np.minimum(prev, mn + P2)
        best[:, 1:] = np.minimum(best[:, 1:], prev[:, :-1] + P1)
        best[:, :-1] = np.minimum(best[:, :-1], prev[:, 1:] + P1)
        return np.where(valid[:, None], cur + best - mn, cur)

    if rx != 0:
        xs = range(W) if rx > 0 else range(W - 1, -1, -1)
        for x in xs:
            px = x - rx
            if px < 0 or px >= W:
                L[:, x] = Ci[:, x]
                continue
            prev = np.zeros((H, D), np.int16)
            valid = np.zeros(H, bool)
            ys = np.arange(H)
            py = ys - ry
            ok = (py >= 0) & (py < H)
            prev[ok] = L[py[ok], px]
            valid[ok] = True
            L[:, x] = step(Ci[:, x], prev, valid)
    else:
        ys = range(H) if ry > 0 else range(H - 1, -1, -1)
        for y in ys:
            py = y - ry
            if py < 0 or py >= H:
                L[y] = Ci[y]
                continue
            L[y] = step(Ci[y], L[py], np.ones(W, bool))
    return L.astype(np.uint8)


def median3(m: np.ndarray) -> np.ndarray:
    H, W = m.shape
    out = m.copy()
    if H >= 3 and W >= 3:
        st = np.stack([m[1 + dy:H - 1 + dy, 1 + dx:W - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], 0)
        out[1:H - 1, 1:W - 1] = np.sort(st, axis=0)[4]
    return out


def compute(left: np.ndarray, right: np.ndarray, D: int = 128, P1: int = 6, P2: int = 96, paths: int = 8, lr_check: bool = True,
            median: bool = True, stages: bool = False):
    H, W = left.shape
    cl, cr = census(left), census(right)
    C = cost_volume(cl, cr, D)
    S = np.zeros((H, W, D), np.uint16)
    Ls = []
    for i in range(paths):
        L = aggregate(C, P1, P2, i)
        Ls.append(L)
        S += L
    dl = S.argmin(axis=2).astype(np.uint8)                     # first minimum
    # right disparity: S(x + d, d) along the diagonal
    big = np.iinfo(np.uint16).max
    Sr = np.full((H, W, D), big, np.uint32)
    for d in range(min(D, W)):
        Sr[:, :W - d, d] = S[:, d:, d]
    dr = Sr.argmin(axis=2).astype(np.uint8)
    if median:
        dl, dr = median3(dl), median3(dr)
    xs = np.arange(W)[None, :] - dl.astype(np.int64)
    ok = xs >= 0
    if lr_check:
        drs = np.take_along_axis(dr, np.clip(xs, 0, W - 1), axis=1).astype(np.int64)
        ok &= np.abs(drs - dl.astype(np.int64)) <= 1
    else:
        ok = np.ones_like(ok)
    disp = np.where(ok, dl.astype(np.float32), np.float32(-1.0)).astype(np.float32)
    if stages:
        return {"census_left": cl, "census_right": cr, "cost": C, "paths": Ls, "S": S, "disparity": disp}
    return disp


def make_stereo(W: int, H: int, seed: int = 0, D: int = 128, n_boxes: int = 4):
    """Synthetic stereo pair (moving_object_detector_amd.synth.make_stereo_images: the input generator lives with the other
    synthetic inputs, outside the oracle)."""
    from moving_object_detector_amd.synth import make_stereo_images
    return make_stereo_images(W, H, seed, D, n_boxes)
