"""Second, independently written restatement of the reference hot path in numpy/scipy.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.cpp header).  "Parity unpinned": the reference has no tests or goldens
and cannot be built here; this file and oracle.cpp were written separately from the reference sources and are
checked against each other (tests/test_oracle_cross.py).  It also generates the committed fixtures in
tests/golden/ (tests/golden/make_golden.py).

Where oracle.cpp follows the reference's *procedure* (serial raster scan + union-find with path halving, std::sort),
this file states the *result* declaratively: vectorised per-pixel formulas, connected
components of the edge graph via scipy.sparse.csgraph, survivors ordered by first_edge_key.  Agreement of the two is
the check.  Reference citations: disparity_image_proc/src/disparity_image_processor.cpp:17-50,
scene_flow_constructor/src/scene_flow_constructor.cpp:65-89,149-212,409-429,
scene_flow_constructor/include/scene_flow_constructor.h:173-249, scene_flow_clusterer/src/clusterer_nodelet.cpp:40-117,
147-219,253-267,354-393.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

f32 = np.float32
f64 = np.float64


def _roundf(a: np.ndarray) -> np.ndarray:
    """std::round on float32: half away from zero (np.round is half-to-even)."""
    a = a.astype(f32)
    t = np.trunc(a)
    frac = np.abs(a - t)                       # exact in binary floating point
    return (t + np.where(frac >= f32(0.5), np.sign(a), f32(0))).astype(f32)


def rotation(q) -> np.ndarray:
    """Quaternion (x,y,z,w) -> 3x3, Eigen::Quaternion::toRotationMatrix operation order, no normalisation."""
    x, y, z, w = (f64(v) for v in q)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]], f64)


def disparity_ok(cam, D):
    """getDisparity range gate (bounds handled by the callers): NaN passes both comparisons."""
    with np.errstate(invalid="ignore"):
        return ~(f32(cam.max_disparity) < D) & ~(f32(cam.min_disparity) > D)


def reproject(cam, D):
    """toPointCloud: (X, Y, Z) float32 planes, NaN where getPoint3D fails."""
    H, W = D.shape
    D = D.astype(f32)
    with np.errstate(all="ignore"):
        ok = disparity_ok(cam, D) & ~(D == f32(0))
        fT = f32(f32(cam.disp_f) * f32(cam.disp_T))
        z = (fT / D).astype(f32)
        rx = (np.arange(W, dtype=f64) - f64(cam.cx) - f64(cam.Tx)) / f64(cam.fx)
        ry = (np.arange(H, dtype=f64) - f64(cam.cy) - f64(cam.Ty)) / f64(cam.fy)
        X = (rx[None, :] * z.astype(f64)).astype(f32)
        Y = (ry[:, None] * z.astype(f64)).astype(f32)
    nan = f32(np.nan)
    return np.where(ok, X, nan), np.where(ok, Y, nan), np.where(ok, z, nan)


def depth_image(cam, D):
    """toDepthImage (disparity_image_processor.cpp:105-120): z of getPoint3D, NaN where it fails."""
    return reproject(cam, D)[2]


def transform_prev(X, Y, Z, t, q):
    """transformPCPreviousToNow: NaN x passes through untouched, else F32(t + R p) with p0 + (p1 + p2) association."""
    R = rotation(q)
    x, y, z = X.astype(f64), Y.astype(f64), Z.astype(f64)
    with np.errstate(all="ignore"):
        out = []
        for i in range(3):
            out.append((f64(t[i]) + (R[i, 0] * x + (R[i, 1] * y + R[i, 2] * z))).astype(f32))
    keep = np.isnan(X)
    return np.where(keep, X, out[0]), np.where(keep, Y, out[1]), np.where(keep, Z, out[2])


def static_flow(cam, Xt, Yt, Zt):
    H, W = Xt.shape
    xs = np.arange(W, dtype=f64)[None, :]
    ys = np.arange(H, dtype=f64)[:, None]
    with np.errstate(all="ignore"):
        u = (f64(cam.fx) * Xt.astype(f64) + f64(cam.Tx)) / Zt.astype(f64) + f64(cam.cx)
        v = (f64(cam.fy) * Yt.astype(f64) + f64(cam.Ty)) / Zt.astype(f64) + f64(cam.cy)
        s0 = (u - xs).astype(f32)
        s1 = (v - ys).astype(f32)
    bad = np.isnan(Xt)
    nan = f32(np.nan)
    return np.where(bad, nan, s0), np.where(bad, nan, s1)


def scene_flow(cam, prm, D_now, D_prev, flow, t, q, dt):
    """construct(): returns dict of float32 planes x,y,z,vx,vy,vz and the static flow (H,W,2)."""
    H, W = D_now.shape
    D_now = D_now.astype(f32)
    D_prev = D_prev.astype(f32)
    Xn, Yn, Zn = reproject(cam, D_now)
    Xp, Yp, Zp = reproject(cam, D_prev)
    Xt, Yt, Zt = transform_prev(Xp, Yp, Zp, t, q)
    s0, s1 = static_flow(cam, Xt, Yt, Zt)
    nan = f32(np.nan)
    with np.errstate(all="ignore"):
        valid_n = ~np.isnan(Xn) & ~np.isinf(Xn)
        f0, f1 = flow[..., 0].astype(f32), flow[..., 1].astype(f32)
        flow_ok = ~np.isnan(f0) & ~np.isnan(f1)
        xs = np.arange(W, dtype=f32)[None, :]
        ys = np.arange(H, dtype=f32)[:, None]
        rx = _roundf((xs - f0).astype(f32))
        ry = _roundf((ys - f1).astype(f32))
        in_int = (rx >= f32(-2147483648.0)) & (rx < f32(2147483648.0)) & (ry >= f32(-2147483648.0)) & (ry < f32(2147483648.0))
        in_img = in_int & (rx >= 0) & (rx < W) & (ry >= 0) & (ry < H)
        px = np.where(in_img, rx, 0).astype(np.int64)
        py = np.where(in_img, ry, 0).astype(np.int64)
        right_now = disparity_ok(cam, D_now) & ~np.isnan(D_now) & ~np.isinf(D_now) & ~(D_now < 0)
        dpw = D_prev[py, px]
        right_prev = in_img & disparity_ok(cam, dpw) & ~np.isnan(dpw) & ~np.isinf(dpw) & ~(dpw < 0)
        Xw, Yw, Zw = Xt[py, px], Yt[py, px], Zt[py, px]
        valid_p = ~np.isnan(Xw) & ~np.isinf(Xw)
        has_v = valid_n & flow_ok & right_now & right_prev & valid_p & ~np.isnan(s0)
        r0 = (f0 - s0).astype(f32)
        r1 = (f1 - s1).astype(f32)
        acc = (f32(0) + (r0 * r0).astype(f32)).astype(f32)
        acc = (acc + (r1 * r1).astype(f32)).astype(f32)
        moving = np.sqrt(acc).astype(f32) >= f32(prm.dynamic_flow_diff)
        vx = ((Xn - Xw).astype(f32).astype(f64) / f64(dt)).astype(f32)
        vy = ((Yn - Yw).astype(f32).astype(f64) / f64(dt)).astype(f32)
        vz = ((Zn - Zw).astype(f32).astype(f64) / f64(dt)).astype(f32)
    zero = f32(0)
    out = {
        "x": np.where(valid_n, Xn, nan), "y": np.where(valid_n, Yn, nan), "z": np.where(valid_n, Zn, nan),
        "vx": np.where(has_v, np.where(moving, vx, zero), nan),
        "vy": np.where(has_v, np.where(moving, vy, zero), nan),
        "vz": np.where(has_v, np.where(moving, vz, zero), nan),
        "static_flow": np.stack([s0, s1], -1),
        "depth": Zn,
    }
    return out


def norm3(vx, vy, vz):
    """Eigen Vector3f::norm(): sqrt(x^2 + (y^2 + z^2)) in float32."""
    with np.errstate(all="ignore"):
        xx = (vx * vx).astype(f32)
        yy = (vy * vy).astype(f32)
        zz = (vz * vz).astype(f32)
        return np.sqrt((xx + (yy + zz).astype(f32)).astype(f32)).astype(f32)


def dynamic_mask(prm, vx, vy, vz):
    with np.errstate(invalid="ignore"):
        return norm3(vx, vy, vz).astype(f64) >= f64(prm.dynamic_speed)


def cluster(prm, x, y, z, vx, vy, vz):
    """clustering() + publishMovingObjects(): returns (labels int32 (H,W), objects list of dict, K)."""
    H, W = z.shape
    N = H * W
    n = int(prm.neighbor_distance)
    dyn = dynamic_mask(prm, vx, vy, vz)
    idx = np.arange(N, dtype=np.int64).reshape(H, W)
    rows, cols = [], []
    has_upleft = np.zeros((H, W), bool)
    for dv in range(-n, 1):
        for du in range(-n, 1):
            if dv == 0 and du == 0:
                continue
            # p = (yy, xx) for yy >= -dv, xx >= -du ; q = p + (du, dv)
            p_sl = (slice(-dv, H), slice(-du, W))
            q_sl = (slice(0, H + dv), slice(0, W + du))
            with np.errstate(invalid="ignore"):
                dd = np.abs((z[p_sl] - z[q_sl]).astype(f32)).astype(f64)
                e = dyn[p_sl] & dyn[q_sl] & ~(dd > f64(prm.depth_diff))
            if e.any():
                rows.append(idx[p_sl][e])
                cols.append(idx[q_sl][e])
                has_upleft[p_sl] |= e
    labels = np.full(N, -1, np.int32)
    objects = []
    if not rows:
        return labels.reshape(H, W), objects, 0
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    g = coo_matrix((np.ones(r.size, np.int8), (r, c)), shape=(N, N))
    _, comp = connected_components(g, directed=False)
    in_edge = np.zeros(N, bool)
    in_edge[r] = True
    in_edge[c] = True
    comp = np.where(in_edge, comp, -1)
    ids, sizes = np.unique(comp[comp >= 0], return_counts=True)
    # first_edge_key: smallest raster index of a member that has an up-left edge
    key = np.full(comp.max() + 1, N, np.int64)
    pe = np.nonzero(has_upleft.ravel())[0]
    np.minimum.at(key, comp[pe], pe)
    keep = [(key[i], i) for i, s in zip(ids, sizes) if s >= prm.cluster_size]
    keep.sort()
    K = len(keep)
    remap = {cid: new for new, (_, cid) in enumerate(keep)}
    for cid, new in remap.items():
        labels[comp == cid] = new
    lab2 = labels.reshape(H, W)
    X, Y, Z, VX, VY, VZ = (a.ravel() for a in (x, y, z, vx, vy, vz))
    nid = 0
    for new in range(K):
        # members in column-major pixel order (clusterMap2IndicesCluster walks u outer, v inner)
        vv, uu = np.nonzero(lab2 == new)
        o = np.lexsort((vv, uu))
        m = (vv[o] * W + uu[o]).astype(np.int64)
        mn = np.array([X[m].min(), Y[m].min(), Z[m].min()], f32)
        mx = np.array([X[m].max(), Y[m].max(), Z[m].max()], f32)
        bbox = (mx - mn).astype(f32)
        center = ((mn + mx).astype(f32) / f32(2)).astype(f32)
        nr = norm3(VX[m], VY[m], VZ[m])
        srt = np.sort(nr)[::-1]
        med = srt[m.size // 2]
        tied = m[nr == med]
        vecs = np.stack([VX[tied], VY[tied], VZ[tied]], -1)
        ambiguous = bool((vecs.view(np.uint32) != vecs[0].view(np.uint32)).any())
        if f64(med) < f64(prm.dynamic_speed):
            continue
        objects.append({"id": nid, "n_points": int(m.size), "center": center.astype(f64), "bounding_box": bbox.astype(f64),
                        "velocity": vecs[0].astype(f64), "velocity_candidates": vecs.astype(f64), "ambiguous": ambiguous,
                        "median_norm": f32(med)})
        nid += 1
    return lab2, objects, K
