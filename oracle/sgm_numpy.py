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
    """Layered synthetic stereo pair with integer disparities: every layer is a random texture translated by its disparity
    between the views (left(x) = T[x], right(x) = T[x + d]); nearer boxes occlude.  Returns left, right (uint8) and the true
    left disparity map."""
    rng = np.random.Generator(np.random.PCG64([0x56D0000 + seed]))

    def texture():
        t = rng.integers(0, 256, size=(H, W + D + 8)).astype(np.float32)
        k = np.array([1, 2, 1], np.float32) / 4
        t = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, t)
        t = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, t)
        return np.clip(t, 0, 255).astype(np.uint8)

    dmax = max(2, min(D - 1, W // 3))
    layers = [(int(rng.integers(1, max(2, dmax // 8))), None, texture())]          # background
    for _ in range(n_boxes):
        bw, bh = int(rng.integers(W // 8, W // 3)), int(rng.integers(H // 6, H // 2))
        x0, y0 = int(rng.integers(0, W - bw)), int(rng.integers(0, H - bh))
        layers.append((int(rng.integers(dmax // 6 + 1, dmax)), (x0, y0, bw, bh), texture()))
    layers.sort(key=lambda l: l[0])                                                 # far to near
    left = np.zeros((H, W), np.uint8)
    right = np.zeros((H, W), np.uint8)
    truth = np.zeros((H, W), np.float32)
    xs = np.arange(W)
    for d, rect, tex in layers:
        ml = np.ones((H, W), bool)
        if rect is not None:
            x0, y0, bw, bh = rect
            ml = np.zeros((H, W), bool)
            ml[y0:y0 + bh, x0:x0 + bw] = True
        mr = np.zeros((H, W), bool)
        mr[:, :W - d] = ml[:, d:]
        if rect is None:
            mr[:] = True
        left = np.where(ml, tex[:, xs], left)
        right = np.where(mr, tex[:, xs + d], right)
        truth = np.where(ml, np.float32(d), truth)
    noise = rng.integers(-2, 3, size=(2, H, W))
    left = np.clip(left.astype(np.int32) + noise[0], 0, 255).astype(np.uint8)
    right = np.clip(right.astype(np.int32) + noise[1], 0, 255).astype(np.uint8)
    return left, right, truth
