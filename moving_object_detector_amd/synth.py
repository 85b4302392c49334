"""Deterministic synthetic stereo+flow inputs for the scene-flow / clustering hot path.

The reference ships no data (no bags, no fixtures), so the workload is build-defined: a ground plane and a
fronto-parallel wall seen by a ZED-like camera, K moving boxes large enough to survive ``cluster_size`` (Clusterer.cfg:8),
two sub-threshold boxes, isolated dynamic pixels, and the invalid-value classes the reference code branches on
(NaN / negative / zero / out-of-range disparity, NaN and huge flow).  What is produced is exactly the four values
``SceneFlowConstructor::construct`` receives (scene_flow_constructor.cpp:91-97): disparity now, disparity previous,
optical flow indexed at the *now* pixel with ``prev = now - flow`` (scene_flow_constructor.h:205-211) and the
previous->now camera transform, plus ``dt``.

Pure numpy, seeded per (config seed, frame index): identical bytes feed the CPU oracle and the HIP path.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# (fx, fy, cx, cy, baseline) per resolution, SURVEY.md §8(d)
_CAMERAS = {
    (640, 480): (525.0, 525.0, 319.5, 239.5, 0.12),
    (1280, 720): (700.0, 700.0, 640.0, 360.0, 0.12),
    (1920, 1080): (1400.0, 1400.0, 960.0, 540.0, 0.12),
}


@dataclass
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    Tx: float = 0.0
    Ty: float = 0.0
    disp_f: float = 0.0
    disp_T: float = 0.12
    min_disparity: float = 0.0
    max_disparity: float = 128.0


@dataclass
class Params:
    """dynamic_reconfigure defaults: SceneFlowConstructor.cfg:8, Clusterer.cfg:8-11."""
    dynamic_flow_diff: int = 5
    cluster_size: int = 2500
    neighbor_distance: int = 4
    depth_diff: float = 0.15
    dynamic_speed: float = 0.3


@dataclass
class Frame:
    disparity_now: np.ndarray      # (H, W) float32
    disparity_prev: np.ndarray     # (H, W) float32
    flow: np.ndarray               # (H, W, 2) float32, x then y
    translation: np.ndarray        # (3,) float64
    quaternion: np.ndarray         # (4,) float64 x, y, z, w
    dt: float
    meta: dict = field(default_factory=dict)


def make_camera(width: int, height: int, style: str = "zed") -> Camera:
    """style "zed": the ZED-like cameras of SURVEY.md §8(d); "kitti": the KITTI-style variant named there (fx = fy = 721.5377,
    cx = 609.5593, baseline 0.5372 m at the native 1242 x 375, scaled to the requested size) — BASELINE.json config 3."""
    if style == "kitti":
        sx, sy = width / 1242.0, height / 375.0
        fx = fy = 721.5377 * sx
        return Camera(width, height, fx, fy, 609.5593 * sx, 172.854 * sy, 0.0, 0.0, np.float32(fx), np.float32(0.5372),
                      np.float32(0.0), np.float32(128.0))
    if style != "zed":
        raise ValueError(f"unknown camera style {style!r}")
    if (width, height) in _CAMERAS:
        fx, fy, cx, cy, base = _CAMERAS[(width, height)]
    else:  # any other size: scale the 1280x720 ZED-like camera
        s = width / 1280.0
        fx = fy = 700.0 * s
        cx, cy, base = (width - 1) / 2.0, (height - 1) / 2.0, 0.12
    return Camera(width, height, fx, fy, cx, cy, 0.0, 0.0, np.float32(fx), np.float32(base), np.float32(0.0),
                  np.float32(128.0))


def _quat_from_yaw_pitch(yaw: float, pitch: float) -> np.ndarray:
    """Camera frame: x right, y down, z forward.  Yaw about y, then pitch about x.  Returns x, y, z, w."""
    cy_, sy_ = np.cos(yaw / 2), np.sin(yaw / 2)
    cp_, sp_ = np.cos(pitch / 2), np.sin(pitch / 2)
    qy = np.array([0.0, sy_, 0.0, cy_])
    qx = np.array([sp_, 0.0, 0.0, cp_])

    def mul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])

    return mul(qy, qx)


def _rot(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _background_depth(cam: Camera, origin: np.ndarray, R: np.ndarray, xs: np.ndarray, ys: np.ndarray,
                      wall_z: float, cam_height: float) -> np.ndarray:
    """Depth (camera z) of the wall/ground seen through pixel (xs, ys) of a camera at `origin` with axes `R`
    (both expressed in the *now* frame, where wall: z = wall_z, ground: y = cam_height)."""
    rx = (xs - cam.cx) / cam.fx
    ry = (ys - cam.cy) / cam.fy
    d = R @ np.stack([rx.ravel(), ry.ravel(), np.ones(rx.size)])
    dx, dy, dz = (d[i].reshape(rx.shape) for i in range(3))
    with np.errstate(divide="ignore", invalid="ignore"):
        s_wall = (wall_z - origin[2]) / dz
        s_ground = np.where(dy > 1e-9, (cam_height - origin[1]) / dy, np.inf)
    s = np.minimum(np.where(s_wall > 0, s_wall, np.inf), np.where(s_ground > 0, s_ground, np.inf))
    return np.clip(s, 2.0, 40.0)


def make_frame(width: int = 1280, height: int = 720, seed: int = 0, frame: int = 0, *, n_objects: int = 6,
               n_small: int = 2, n_isolated: int = 24, dt: float = 0.1, quantize: bool = True,
               invalid: bool = True, camera: str = "zed") -> tuple[Camera, Frame]:
    """One stereo pair (now + previous disparity), its flow and ego-motion.

    Object depth 4-9 m, speed 1-2 m/s and dt = 0.1 s (the KITTI-style value of SURVEY.md §8(d)) are chosen so that the
    image-plane residual exceeds the reference's default ``dynamic_flow_diff`` of 5 px; slower / farther objects are
    (faithfully) not detected by the reference at its defaults.
    """
    cam = make_camera(width, height, camera)
    W, H = width, height
    rng = np.random.Generator(np.random.PCG64([0x5EED0000 + seed, frame]))
    sc = W / 1280.0                                        # size scale relative to the 720p layout
    f, base = float(cam.disp_f), float(cam.disp_T)

    # ego-motion previous -> now: P_now = R P_prev + t (jittered around SURVEY's nominal values)
    t = np.array([0.01, 0.0, 0.08]) * (1.0 + 0.2 * rng.uniform(-1, 1, 3))
    yaw = np.deg2rad(0.4) * (1.0 + 0.2 * rng.uniform(-1, 1))
    pitch = np.deg2rad(0.1) * (1.0 + 0.2 * rng.uniform(-1, 1))
    q = _quat_from_yaw_pitch(yaw, pitch)
    R = _rot(q)

    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    wall_z, cam_h = 25.0 + 10.0 * rng.uniform(), 1.5

    # --- now frame: background depth, then paint objects nearest-first-wins via a z-buffer -----------------------
    z_now = _background_depth(cam, np.zeros(3), np.eye(3), xs, ys, wall_z, cam_h)
    obj_id = np.full((H, W), -1, np.int32)
    vel = np.zeros((H, W, 3))
    # previous frame is seen from origin -R^T t with axes R^T (in now coordinates: centre = t... see below)
    # P_now = R P_prev + t  =>  prev camera centre in now coords is t, prev axes are the columns of R.
    z_prev = _background_depth(cam, t, R, xs, ys, wall_z, cam_h)
    prev_obj = np.full((H, W), -1, np.int32)

    boxes = []
    for k in range(n_objects + n_small):
        small = k >= n_objects
        if small:
            bw, bh = int(rng.integers(24, 44) * sc), int(rng.integers(24, 44) * sc)
        else:
            bw, bh = int(rng.integers(90, 261) * sc), int(rng.integers(120, 401) * sc)
        bw, bh = max(bw, 3), max(bh, 3)
        zc = rng.uniform(4.0, 9.0)
        x0 = int(rng.integers(8, max(9, W - bw - 8)))
        y0 = int(rng.integers(8, max(9, H - bh - 8)))
        speed = rng.uniform(1.0, 2.0)
        ang = rng.uniform(0, 2 * np.pi)
        v = np.array([speed * np.cos(ang), 0.15 * speed * rng.uniform(-1, 1), 0.5 * speed * np.sin(ang)])
        slant = rng.uniform(-0.3, 0.3) / max(bw, 1)        # depth gradient across the box [m/px]
        boxes.append((x0, y0, bw, bh, zc, v, slant))
    order = np.argsort([-b[4] for b in boxes])             # far to near, near overwrites
    for k in order:
        x0, y0, bw, bh, zc, v, slant = boxes[k]
        sl = (slice(y0, y0 + bh), slice(x0, x0 + bw))
        zz = zc + slant * (xs[sl] - (x0 + bw / 2.0))
        closer = zz < z_now[sl]
        z_now[sl] = np.where(closer, zz, z_now[sl])
        obj_id[sl] = np.where(closer, k, obj_id[sl])
        vel[sl] = np.where(closer[..., None], v, vel[sl])
        # the same box one frame earlier, as seen by the previous camera (kept fronto-parallel)
        c_now = np.array([(x0 + bw / 2.0 - cam.cx) / cam.fx * zc, (y0 + bh / 2.0 - cam.cy) / cam.fy * zc, zc])
        c_prev = R.T @ (c_now - v * dt - t)
        pw, ph = bw * zc / c_prev[2], bh * zc / c_prev[2]
        px0 = int(round(cam.fx * c_prev[0] / c_prev[2] + cam.cx - pw / 2.0))
        py0 = int(round(cam.fy * c_prev[1] / c_prev[2] + cam.cy - ph / 2.0))
        xa, xb = max(px0, 0), min(px0 + int(round(pw)), W)
        ya, yb = max(py0, 0), min(py0 + int(round(ph)), H)
        if xa < xb and ya < yb:
            psl = (slice(ya, yb), slice(xa, xb))
            pz = c_prev[2] + slant * (xs[psl] - (px0 + pw / 2.0))
            pcl = pz < z_prev[psl]
            z_prev[psl] = np.where(pcl, pz, z_prev[psl])
            prev_obj[psl] = np.where(pcl, k, prev_obj[psl])

    # --- flow at the now pixel: where was this surface point one frame ago? ---------------------------------------
    Pn = np.stack([(xs - cam.cx) / cam.fx * z_now, (ys - cam.cy) / cam.fy * z_now, z_now], -1)
    Pp = (Pn - vel * dt - t) @ R                            # R^T (P - v dt - t), row-vector form
    with np.errstate(divide="ignore", invalid="ignore"):
        xp = cam.fx * Pp[..., 0] / Pp[..., 2] + cam.cx
        yp = cam.fy * Pp[..., 1] / Pp[..., 2] + cam.cy
    flow = np.stack([xs - xp, ys - yp], -1)
    flow += rng.uniform(-0.3, 0.3, flow.shape)

    d_now = f * base / z_now
    d_prev = f * base / z_prev
    if quantize:                                            # quarter-pixel SGM-like quantisation
        d_now = np.round(d_now * 4.0) / 4.0
        d_prev = np.round(d_prev * 4.0) / 4.0
    d_now = d_now.astype(np.float32)
    d_prev = d_prev.astype(np.float32)
    flow = flow.astype(np.float32)

    # isolated dynamic pixels: a flow outlier of 10-30 px on a background pixel
    n_iso = 0
    for _ in range(n_isolated):
        mg = min(40, W // 8, H // 8)
        x, y = int(rng.integers(mg, W - mg)), int(rng.integers(mg, H - mg))
        if obj_id[y, x] >= 0:
            continue
        a = rng.uniform(0, 2 * np.pi)
        r = rng.uniform(10, 30)
        flow[y, x, 0] += np.float32(r * np.cos(a))
        flow[y, x, 1] += np.float32(r * np.sin(a))
        n_iso += 1

    if invalid:
        for d in (d_now, d_prev):
            u = rng.random((H, W))
            d[u < 0.03] = np.nan
            d[(u >= 0.03) & (u < 0.04)] = -1.0
            d[(u >= 0.04) & (u < 0.05)] = 0.0
            d[(u >= 0.05) & (u < 0.06)] = 200.0            # > max_disparity
        u = rng.random((H, W))
        flow[u < 0.01] = np.nan
        huge = (u >= 0.01) & (u < 0.011)
        flow[huge] = np.where(rng.random((int(huge.sum()), 2)) < 0.5, np.float32(-1e6), np.float32(1e6))

    meta = {"n_boxes": n_objects, "n_small": n_small, "n_isolated": n_iso,
            "object_pixels": int((obj_id >= 0).sum()), "wall_z": wall_z}
    return cam, Frame(d_now, d_prev, flow, t.astype(np.float64), q.astype(np.float64), float(dt), meta)


def make_batch(width: int, height: int, frames: int, seed: int = 0, first_frame: int = 0, **kw):
    """`frames` independent pairs stacked: disparity_now/prev (F,H,W), flow (F,H,W,2), t (F,3), q (F,4), dt (F,)."""
    cam = make_camera(width, height, kw.get("camera", "zed"))
    dn = np.empty((frames, height, width), np.float32)
    dp = np.empty((frames, height, width), np.float32)
    fl = np.empty((frames, height, width, 2), np.float32)
    ts = np.empty((frames, 3), np.float64)
    qs = np.empty((frames, 4), np.float64)
    dts = np.empty((frames,), np.float64)
    for i in range(frames):
        _, fr = make_frame(width, height, seed, first_frame + i, **kw)
        dn[i], dp[i], fl[i], ts[i], qs[i], dts[i] = fr.disparity_now, fr.disparity_prev, fr.flow, fr.translation, fr.quaternion, fr.dt
    return cam, {"disparity_now": dn, "disparity_prev": dp, "flow": fl, "t": ts, "q": qs, "dt": dts}


# ---- a proper stream: frame t's "now" disparity IS frame t+1's "previous" disparity --------------------------------------
def _ego_step(seed: int, k: int):
    """previous(k-1) -> now(k) camera motion of stream frame k, seeded per frame index (so any sub-range of the stream can be
    generated on its own, e.g. one rank's shard)."""
    rng = np.random.Generator(np.random.PCG64([0x5EED1000 + seed, k, 1]))
    t = np.array([0.01, 0.0, 0.08]) * (1.0 + 0.2 * rng.uniform(-1, 1, 3))
    yaw = np.deg2rad(0.4) * (1.0 + 0.2 * rng.uniform(-1, 1)) * (1.0 if (k // 32) % 2 == 0 else -1.0)   # weave, do not circle
    pitch = np.deg2rad(0.1) * rng.uniform(-1, 1)
    q = _quat_from_yaw_pitch(yaw, pitch)
    return t, q, _rot(q)


def _tri(k: float, period: int) -> float:
    """Triangle wave in [-1, 1] with the given period (object paths go back and forth and stay in view)."""
    u = (k / period) % 1.0
    return 4.0 * u - 1.0 if u < 0.5 else 3.0 - 4.0 * u


class _Scene:
    """World of one stream: wall + ground + moving boxes, all a function of (seed, size) only.  World = camera frame 0."""

    def __init__(self, cam: Camera, seed: int, n_objects: int, n_small: int, depth=(4.0, 9.0), speed=(1.0, 2.0)):
        self.cam = cam
        W, H = cam.width, cam.height
        rng = np.random.Generator(np.random.PCG64([0x5EED1000 + seed, 0, 2]))
        self.wall_z, self.cam_h = 60.0 + 10.0 * rng.uniform(), 1.5
        sc = W / 1280.0
        self.boxes = []
        for k in range(n_objects + n_small):
            small = k >= n_objects
            if small:
                bw, bh = int(rng.integers(24, 44) * sc), int(rng.integers(24, 44) * sc)
            else:
                bw, bh = int(rng.integers(90, 261) * sc), int(rng.integers(120, 401) * sc)
            bw, bh = max(bw, 3), max(bh, 3)
            zc = rng.uniform(depth[0], depth[1])
            x0 = int(rng.integers(8, max(9, W - bw - 8)))
            y0 = int(rng.integers(8, max(9, H - bh - 8)))
            spd = rng.uniform(speed[0], speed[1])
            ang = rng.uniform(0, 2 * np.pi)
            v = np.array([spd * np.cos(ang), 0.15 * spd * rng.uniform(-1, 1), 0.5 * spd * np.sin(ang)])
            slant = rng.uniform(-0.3, 0.3)                 # depth change across the box [m]
            # metric size and centre RELATIVE TO THE CAMERA (the boxes ride along with the vehicle, like traffic ahead of it)
            wm, hm = bw * zc / cam.fx, bh * zc / cam.fy
            c0 = np.array([(x0 + bw / 2.0 - cam.cx) / cam.fx * zc, (y0 + bh / 2.0 - cam.cy) / cam.fy * zc, zc])
            period = int(rng.integers(6, 14))
            self.boxes.append((c0, v, wm, hm, slant, period, float(rng.uniform(0, 1))))

    def box_centre(self, j: int, k: int, dt: float) -> np.ndarray:
        """Centre of box j in the camera frame of stream index k: a constant-speed back-and-forth path around c0, so the
        camera-relative velocity between consecutive frames is +-v (1-2 m/s) — what the residual test sees."""
        c0, v, _, _, _, period, ph = self.boxes[j]
        return c0 + v * dt * (period / 4.0) * _tri(k + ph * period, period)


def _render(scene: _Scene, A: np.ndarray, b: np.ndarray, k: int, dt: float, xs, ys):
    """Depth and object id seen by the camera of stream index k (P_k = A P_world + b)."""
    cam = scene.cam
    H, W = xs.shape
    z = _background_depth(cam, -A.T @ b, A.T, xs, ys, scene.wall_z, scene.cam_h)
    obj = np.full((H, W), -1, np.int32)
    cents = [scene.box_centre(j, k, dt) for j in range(len(scene.boxes))]
    for j in np.argsort([-c[2] for c in cents]):           # far to near, near overwrites
        c = cents[j]
        _, _, wm, hm, slant, _, _ = scene.boxes[j]
        pw, ph = wm * cam.fx / c[2], hm * cam.fy / c[2]
        px0 = int(round(cam.fx * c[0] / c[2] + cam.cx - pw / 2.0))
        py0 = int(round(cam.fy * c[1] / c[2] + cam.cy - ph / 2.0))
        xa, xb = max(px0, 0), min(px0 + int(round(pw)), W)
        ya, yb = max(py0, 0), min(py0 + int(round(ph)), H)
        if xa >= xb or ya >= yb:
            continue
        sl = (slice(ya, yb), slice(xa, xb))
        zz = c[2] + slant * (xs[sl] - (px0 + pw / 2.0)) / max(pw, 1.0)
        closer = zz < z[sl]
        z[sl] = np.where(closer, zz, z[sl])
        obj[sl] = np.where(closer, j, obj[sl])
    return z, obj


NOMINAL = {"depth": (4.0, 15.0), "speed": (0.5, 2.0), "dt": 1.0 / 15.0}   # SURVEY.md section 8(d)'s nominal values


def make_sequence(width: int, height: int, frames: int, seed: int = 0, first: int = 0, *, n_objects: int = 6, n_small: int = 2,
                  n_isolated: int = 24, dt: float = 0.1, quantize: bool = True, invalid: bool = True, camera: str = "zed",
                  depth=(4.0, 9.0), speed=(1.0, 2.0)):
    """A stream as the reference's stereoCallback sees it (scene_flow_constructor.cpp:364-399): disparity planes D[first ..
    first + frames] (frames + 1 planes: frame t pairs previous = D[t] with now = D[t + 1], as `disparity_previous_ =
    disparity_now_` does, :397-398), and per frame t the flow at the now pixel, the previous->now transform and dt.

    Every plane / frame is a function of (seed, its own stream index) only, so a rank can generate exactly its shard
    (contiguous chunk + one disparity plane of halo) and get the bytes a single process would.
    Returns (camera, {"disparity": (frames+1,H,W), "flow": (frames,H,W,2), "t": (frames,3), "q": (frames,4), "dt": (frames,)}).
    """
    cam = make_camera(width, height, camera)
    W, H = width, height
    scene = _Scene(cam, seed, n_objects, n_small, depth, speed)
    f, base = float(cam.disp_f), float(cam.disp_T)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    # pose of camera index i (plane D[i]): P_i = A_i P_world + b_i, chained from index 0
    A, b = np.eye(3), np.zeros(3)
    for i in range(1, first + 1):
        t_, _, R_ = _ego_step(seed, i)
        A, b = R_ @ A, R_ @ b + t_

    def plane(i, A, b):
        z, obj = _render(scene, A, b, i, dt, xs, ys)
        d = f * base / z
        if quantize:
            d = np.round(d * 4.0) / 4.0
        d = d.astype(np.float32)
        if invalid:
            u = np.random.Generator(np.random.PCG64([0x5EED1000 + seed, i, 3])).random((H, W))
            d[u < 0.03] = np.nan
            d[(u >= 0.03) & (u < 0.04)] = -1.0
            d[(u >= 0.04) & (u < 0.05)] = 0.0
            d[(u >= 0.05) & (u < 0.06)] = 200.0
        return d, z, obj

    D = np.empty((frames + 1, H, W), np.float32)
    fl = np.empty((frames, H, W, 2), np.float32)
    ts = np.empty((frames, 3), np.float64)
    qs = np.empty((frames, 4), np.float64)
    dts = np.full((frames,), float(dt), np.float64)
    D[0], _, _ = plane(first, A, b)
    for n in range(frames):
        i = first + n + 1                                  # stream index of the "now" plane of frame first + n
        t_, q_, R_ = _ego_step(seed, i)
        A, b = R_ @ A, R_ @ b + t_
        D[n + 1], z_now, obj = plane(i, A, b)
        # flow at the now pixel: where was this surface point one frame ago, in the previous camera?
        Pn = np.stack([(xs - cam.cx) / cam.fx * z_now, (ys - cam.cy) / cam.fy * z_now, z_now], -1)
        disp = np.zeros((H, W, 3))                         # camera-relative displacement of the surface since the last frame
        for j in range(len(scene.boxes)):
            disp[obj == j] = scene.box_centre(j, i, dt) - scene.box_centre(j, i - 1, dt)
        # background: P_prev = R^T (P_now - t); a box rides with the camera: its point was at (P_now - disp) in the previous
        # camera's own coordinates
        Pp = np.where((obj >= 0)[..., None], Pn - disp, (Pn - t_) @ R_)
        with np.errstate(divide="ignore", invalid="ignore"):
            xp = cam.fx * Pp[..., 0] / Pp[..., 2] + cam.cx
            yp = cam.fy * Pp[..., 1] / Pp[..., 2] + cam.cy
        rng = np.random.Generator(np.random.PCG64([0x5EED1000 + seed, i, 4]))
        flow = np.stack([xs - xp, ys - yp], -1) + rng.uniform(-0.3, 0.3, (H, W, 2))
        flow = flow.astype(np.float32)
        mg = min(40, W // 8, H // 8)
        for _ in range(n_isolated):                        # isolated dynamic pixels: flow outliers on the background
            x, y = int(rng.integers(mg, W - mg)), int(rng.integers(mg, H - mg))
            a, r = rng.uniform(0, 2 * np.pi), rng.uniform(10, 30)
            if obj[y, x] < 0:
                flow[y, x, 0] += np.float32(r * np.cos(a))
                flow[y, x, 1] += np.float32(r * np.sin(a))
        if invalid:
            u = rng.random((H, W))
            flow[u < 0.01] = np.nan
            huge = (u >= 0.01) & (u < 0.011)
            flow[huge] = np.where(rng.random((int(huge.sum()), 2)) < 0.5, np.float32(-1e6), np.float32(1e6))
        fl[n], ts[n], qs[n] = flow, t_, q_
    return cam, {"disparity": D, "flow": fl, "t": ts, "q": qs, "dt": dts}


# ---- stereo IMAGES for the on-GPU disparity estimator (BASELINE config 5) ------------------------------------
def make_stereo_images(W: int, H: int, seed: int = 0, D: int = 128, n_boxes: int = 4):
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


def make_box_flow(truth: np.ndarray, shift: float = 24.0) -> np.ndarray:
    """Optical flow (H, W, 2) for a stereo pair of make_stereo_images seen twice by a camera that stands still: zero on the
    background, (-shift, 0) on the boxes (every layer nearer than the background), i.e. a box pixel was `shift` pixels to the
    right in the previous frame.  With identity ego-motion the residual against the static flow is `shift` px on the boxes
    (dynamic at the reference's default dynamic_flow_diff = 5) and 0 elsewhere, so the clusterer finds the boxes."""
    H, W = truth.shape
    flow = np.zeros((H, W, 2), np.float32)
    flow[truth > truth.min(), 0] = -np.float32(shift)
    return flow
