"""Deterministic synthetic LiDAR scans (no datasets are reachable from the build or GPU boxes).

SURVEY.md §8(d): scene = ground plane + axis-aligned boxes/walls + vertical cylinders within 60 m; a spinning
multi-beam sensor is ray-cast against it, range noise N(0, 0.02 m), max range 100 m, rays without a hit dropped.
Sensor models follow the devices hdl_graph_slam's launch files target (VLP-16 / HDL-32E: launch/hdl_graph_slam*.launch,
64-beam KITTI: launch/hdl_graph_slam_kitti.launch).  Output clouds are pcl::PointXYZI-shaped records
(x, y, z, 1.0 | intensity, 0, 0, 0 -> 32 bytes, the layout KeyFrame::cloud uses, include/hdl_graph_slam/keyframe.hpp:42).
"""
from __future__ import annotations

import dataclasses
import hashlib
import os
import tempfile

import numpy as np

POINT_XYZI_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("intensity", "<f4"), ("pad", "<f4", (3,))])
assert POINT_XYZI_DTYPE.itemsize == 32

SENSORS = {
    # name: (n_beams, elevation_min_deg, elevation_max_deg, azimuth_steps, mount height [m])
    "VLP-16": (16, -15.0, 15.0, 1800, 1.2),
    "HDL-32E": (32, -30.67, 10.67, 2170, 1.5),
    "HDL-64E": (64, -24.8, 2.0, 1875, 1.73),
}


@dataclasses.dataclass
class Scene:
    box_min: np.ndarray  # [nb, 3]
    box_max: np.ndarray  # [nb, 3]
    cyl_xy: np.ndarray   # [nc, 2]
    cyl_r: np.ndarray    # [nc]
    cyl_h: np.ndarray    # [nc]


def make_scene(scene_seed: int, n_boxes: int = 80, n_cyl: int = 20, extent: float = 60.0) -> Scene:
    rng = np.random.default_rng(10_000 + scene_seed)
    ctr = rng.uniform(-extent, extent, size=(n_boxes, 2))
    # keep a free corridor along x around y=0 so a vehicle trajectory does not start inside a box
    ctr[:, 1] = np.where(np.abs(ctr[:, 1]) < 6.0, np.sign(ctr[:, 1] + 1e-9) * (6.0 + np.abs(ctr[:, 1])), ctr[:, 1])
    half = np.stack([rng.uniform(0.5, 6.0, n_boxes), rng.uniform(0.5, 6.0, n_boxes)], axis=1)
    # a third of the boxes are thin long walls
    wall = rng.random(n_boxes) < 0.33
    axis = rng.integers(0, 2, n_boxes)
    half[wall, 0] = np.where(axis[wall] == 0, rng.uniform(4.0, 15.0, wall.sum()), 0.15)
    half[wall, 1] = np.where(axis[wall] == 0, 0.15, rng.uniform(4.0, 15.0, wall.sum()))
    height = rng.uniform(1.0, 8.0, n_boxes)
    box_min = np.concatenate([ctr - half, np.zeros((n_boxes, 1))], axis=1)
    box_max = np.concatenate([ctr + half, height[:, None]], axis=1)
    cyl_xy = rng.uniform(-extent, extent, size=(n_cyl, 2))
    cyl_xy[:, 1] = np.where(np.abs(cyl_xy[:, 1]) < 4.0, np.sign(cyl_xy[:, 1] + 1e-9) * (4.0 + np.abs(cyl_xy[:, 1])), cyl_xy[:, 1])
    cyl_r = rng.uniform(0.15, 0.6, n_cyl)
    cyl_h = rng.uniform(3.0, 10.0, n_cyl)
    return Scene(box_min.astype(np.float64), box_max.astype(np.float64), cyl_xy, cyl_r, cyl_h)


def pose_matrix(xyz, rpy) -> np.ndarray:
    """4x4 (float64) = Trans(xyz) * Rz(yaw) * Ry(pitch) * Rx(roll)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = xyz
    return T


def _ray_dirs(sensor: str) -> np.ndarray:
    nb, e0, e1, naz, _ = SENSORS[sensor]
    elev = np.deg2rad(np.linspace(e0, e1, nb))
    az = np.linspace(0.0, 2 * np.pi, naz, endpoint=False)
    ce, se = np.cos(elev), np.sin(elev)
    # azimuth-major firing order (all beams of one azimuth, then the next), like a spinning sensor's packets
    d = np.stack([np.outer(np.cos(az), ce), np.outer(np.sin(az), ce), np.outer(np.ones_like(az), se)], axis=-1)
    return d.reshape(-1, 3)


def _cast(scene: Scene, o: np.ndarray, d: np.ndarray, max_range: float) -> np.ndarray:
    """Nearest hit distance per ray (inf if none). o: [3], d: [R,3] unit vectors (world frame)."""
    R = d.shape[0]
    t_best = np.full(R, np.inf)
    # ground z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = -o[2] / d[:, 2]
    tg = np.where((d[:, 2] < 0) & (tg > 0), tg, np.inf)
    t_best = np.minimum(t_best, tg)
    # boxes: slab test, chunked over rays
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    chunk = 16384
    for s in range(0, R, chunk):
        e = min(R, s + chunk)
        iv = inv[s:e, None, :]                                   # [r,1,3]
        t0 = (scene.box_min[None, :, :] - o[None, None, :]) * iv  # [r,nb,3]
        t1 = (scene.box_max[None, :, :] - o[None, None, :]) * iv
        tn = np.minimum(t0, t1).max(axis=2)
        tf = np.maximum(t0, t1).min(axis=2)
        hit = (tf >= tn) & (tf > 0)
        t = np.where(tn > 0, tn, tf)  # origin inside a box -> exit point
        t = np.where(hit, t, np.inf).min(axis=1)
        t_best[s:e] = np.minimum(t_best[s:e], t)
        # cylinders (vertical, base z=0, height h): quadratic in xy
        dx, dy = d[s:e, 0:1], d[s:e, 1:2]
        ox = o[0] - scene.cyl_xy[None, :, 0]
        oy = o[1] - scene.cyl_xy[None, :, 1]
        a = dx * dx + dy * dy
        b = 2 * (ox * dx + oy * dy)
        c = ox * ox + oy * oy - scene.cyl_r[None, :] ** 2
        disc = b * b - 4 * a * c
        with np.errstate(divide="ignore", invalid="ignore"):
            sq = np.sqrt(np.where(disc >= 0, disc, np.nan))
            tc = (-b - sq) / (2 * a)
        z = o[2] + tc * d[s:e, 2:3]
        ok = (disc >= 0) & (tc > 0) & (z >= 0) & (z <= scene.cyl_h[None, :])
        tc = np.where(ok, tc, np.inf).min(axis=1)
        t_best[s:e] = np.minimum(t_best[s:e], tc)
    t_best[t_best > max_range] = np.inf
    return t_best


def scan(scene: Scene, sensor: str, pose: np.ndarray, noise_seed: int, noise_sigma: float = 0.02, max_range: float = 100.0,
         as_xyzi: bool = True) -> np.ndarray:
    """Ray-cast one revolution from `pose` (4x4 vehicle pose in the world; the sensor sits `mount height` above it).

    Returns the hits in the SENSOR frame, in firing order, as PointXYZI records (or [n,3] float32)."""
    # Ray casting is 1-2 s per 64-beam revolution on the host; repeated runs of the benchmarks / A-B scripts over the same seeded scenes (a GPU
    # visit runs bench.py a dozen times) read the revolution back from a per-user scratch directory instead (_scan_cache_root).  HGS_SCAN_CACHE=<dir> (empty: off); the key is
    # every input of this function plus this file's own bytes, so an edit to the simulator never meets a stale scan.
    cache = _scan_cache_path(scene, sensor, pose, noise_seed, noise_sigma, max_range, as_xyzi)
    if cache and os.path.exists(cache):
        try:
            return np.load(cache)
        except Exception:  # noqa: BLE001 - a truncated file of an interrupted run: cast again
            pass
    out = _scan_uncached(scene, sensor, pose, noise_seed, noise_sigma, max_range, as_xyzi)
    if cache:
        try:
            tmp = f"{cache}.{os.getpid()}.tmp.npy"
            np.save(tmp, out)
            os.replace(tmp, cache)
        except OSError:
            pass
    return out


_SELF_DIGEST = None  # sha256 of this file, computed once per process (the cache key's "simulator version")


def _scan_cache_root():
    """HGS_SCAN_CACHE=<dir> (empty: off).  Default: a PER-USER directory (~/.cache/hgs_scan_cache, mode 0700, owned by this user) — not a shared,
    predictable path under /tmp, where another user could plant files the benchmarks would load as ray-cast scans."""
    root = os.environ.get("HGS_SCAN_CACHE")
    explicit = root is not None
    if root is None:
        root = os.path.join(os.path.expanduser("~"), ".cache", "hgs_scan_cache")
    if not root:
        return None
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
        if not explicit:
            st = os.stat(root)
            if st.st_uid != os.getuid() or (st.st_mode & 0o022):
                return None
    except OSError:
        return None
    return root


def _scan_cache_path(scene, sensor, pose, noise_seed, noise_sigma, max_range, as_xyzi):
    global _SELF_DIGEST
    root = _scan_cache_root()
    if not root:
        return None
    if _SELF_DIGEST is None:
        with open(os.path.abspath(__file__), "rb") as fh:
            _SELF_DIGEST = hashlib.sha256(fh.read()).digest()
    h = hashlib.sha256()
    h.update(_SELF_DIGEST)
    for f in dataclasses.fields(scene):
        v = getattr(scene, f.name)
        h.update(np.ascontiguousarray(v).tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
    h.update(repr((sensor, int(noise_seed), float(noise_sigma), float(max_range), bool(as_xyzi))).encode())
    h.update(np.ascontiguousarray(pose, dtype=np.float64).tobytes())
    return os.path.join(root, h.hexdigest()[:32] + ".npy")


def _scan_uncached(scene: Scene, sensor: str, pose: np.ndarray, noise_seed: int, noise_sigma: float, max_range: float, as_xyzi: bool) -> np.ndarray:
    dirs_s = _ray_dirs(sensor)
    T = pose.copy()
    T[2, 3] += SENSORS[sensor][4]
    o = T[:3, 3]
    d_w = dirs_s @ T[:3, :3].T
    t = _cast(scene, o, d_w, max_range)
    rng = np.random.default_rng(noise_seed)
    t = t + rng.normal(0.0, noise_sigma, size=t.shape)
    keep = np.isfinite(t) & (t > 0.5)
    pts = (dirs_s[keep] * t[keep, None]).astype(np.float32)
    if not as_xyzi:
        return pts
    return to_xyzi(pts, intensity=(t[keep] % 1.0).astype(np.float32))


def to_xyzi(xyz: np.ndarray, intensity=None) -> np.ndarray:
    out = np.zeros(xyz.shape[0], dtype=POINT_XYZI_DTYPE)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out["w"] = 1.0
    if intensity is not None:
        out["intensity"] = intensity
    return out


def xyz_of(cloud: np.ndarray) -> np.ndarray:
    if cloud.dtype == POINT_XYZI_DTYPE:
        return np.stack([cloud["x"], cloud["y"], cloud["z"]], axis=1)
    return np.asarray(cloud, dtype=np.float32).reshape(-1, cloud.shape[-1])[:, :3]


def voxel_downsample(cloud: np.ndarray, leaf: float) -> np.ndarray:
    """Centroid voxel-grid downsample (the prefilter of apps/prefiltering_nodelet.cpp:51-66, pcl::VoxelGrid)."""
    xyz = xyz_of(cloud).astype(np.float64)
    ijk = np.floor(xyz / leaf).astype(np.int64)
    ijk -= ijk.min(axis=0)
    dims = ijk.max(axis=0) + 1
    key = (ijk[:, 2] * dims[1] + ijk[:, 1]) * dims[0] + ijk[:, 0]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    starts = np.flatnonzero(np.concatenate([[True], key_s[1:] != key_s[:-1]]))
    counts = np.diff(np.concatenate([starts, [len(key_s)]]))
    sums = np.add.reduceat(xyz[order], starts, axis=0)
    return to_xyzi((sums / counts[:, None]).astype(np.float32))


def random_relative_pose(pose_seed: int, trans=(0.2, 1.5), yaw_deg=3.0, rp_deg=0.5) -> np.ndarray:
    """Ground-truth relative motion between two scans: mostly along x."""
    rng = np.random.default_rng(pose_seed)
    dist = rng.uniform(*trans)
    head = np.deg2rad(rng.uniform(-10, 10))
    xyz = np.array([dist * np.cos(head), dist * np.sin(head), rng.uniform(-0.02, 0.02)])
    rpy = np.deg2rad([rng.uniform(-rp_deg, rp_deg), rng.uniform(-rp_deg, rp_deg), rng.uniform(-yaw_deg, yaw_deg)])
    return pose_matrix(xyz, rpy)


def make_pair(sensor: str, scene_seed: int, downsample: float | None = None):
    """(target_cloud, source_cloud, T_gt) with T_gt mapping source-frame points into the target frame."""
    scene = make_scene(scene_seed)
    pose_t = pose_matrix([0.0, 0.0, 0.0], [0.0, 0.0, 0.0])
    rel = random_relative_pose(2000 + scene_seed)
    pose_s = pose_t @ rel
    tgt = scan(scene, sensor, pose_t, 1000 + scene_seed)
    src = scan(scene, sensor, pose_s, 3000 + scene_seed)
    if downsample:
        tgt, src = voxel_downsample(tgt, downsample), voxel_downsample(src, downsample)
    return tgt, src, rel


def dense_surface_cloud(scene: Scene, n_points: int, sample_seed: int, noise_sigma: float = 0.02, extent: float = 60.0) -> np.ndarray:
    """Config 5: area-uniform samples on the scene surfaces (ground + box faces + cylinder sides)."""
    rng = np.random.default_rng(sample_seed)
    ext = scene.box_max - scene.box_min
    areas_box = 2 * (ext[:, 0] * ext[:, 2] + ext[:, 1] * ext[:, 2]) + ext[:, 0] * ext[:, 1]
    areas_cyl = 2 * np.pi * scene.cyl_r * scene.cyl_h
    area_ground = (2 * extent) ** 2
    total = area_ground + areas_box.sum() + areas_cyl.sum()
    n_ground = int(n_points * area_ground / total)
    n_cyl = int(n_points * areas_cyl.sum() / total)
    n_box = n_points - n_ground - n_cyl
    out = [np.stack([rng.uniform(-extent, extent, n_ground), rng.uniform(-extent, extent, n_ground), np.zeros(n_ground)], axis=1)]
    bi = rng.choice(len(areas_box), size=n_box, p=areas_box / areas_box.sum())
    u, v, f = rng.random(n_box), rng.random(n_box), rng.random(n_box)
    e = ext[bi]
    a_xz, a_yz, a_top = e[:, 0] * e[:, 2], e[:, 1] * e[:, 2], e[:, 0] * e[:, 1]
    tot = 2 * a_xz + 2 * a_yz + a_top
    sel = f * tot
    p = np.empty((n_box, 3))
    m0 = sel < a_xz
    m1 = ~m0 & (sel < 2 * a_xz)
    m2 = ~m0 & ~m1 & (sel < 2 * a_xz + a_yz)
    m3 = ~m0 & ~m1 & ~m2 & (sel < 2 * a_xz + 2 * a_yz)
    m4 = ~(m0 | m1 | m2 | m3)
    lo = scene.box_min[bi]
    p[:, 0] = lo[:, 0] + u * e[:, 0]
    p[:, 1] = lo[:, 1] + v * e[:, 1]
    p[:, 2] = lo[:, 2] + v * e[:, 2]
    p[m0, 1] = lo[m0, 1]
    p[m1, 1] = lo[m1, 1] + e[m1, 1]
    p[m2, 0] = lo[m2, 0]
    p[m2, 1] = lo[m2, 1] + u[m2] * e[m2, 1]
    p[m3, 0] = lo[m3, 0] + e[m3, 0]
    p[m3, 1] = lo[m3, 1] + u[m3] * e[m3, 1]
    p[m4, 2] = lo[m4, 2] + e[m4, 2]
    out.append(p)
    ci = rng.choice(len(areas_cyl), size=n_cyl, p=areas_cyl / areas_cyl.sum())
    th = rng.uniform(0, 2 * np.pi, n_cyl)
    out.append(np.stack([scene.cyl_xy[ci, 0] + scene.cyl_r[ci] * np.cos(th), scene.cyl_xy[ci, 1] + scene.cyl_r[ci] * np.sin(th),
                         rng.random(n_cyl) * scene.cyl_h[ci]], axis=1))
    pts = np.concatenate(out, axis=0)
    pts += rng.normal(0, noise_sigma, size=pts.shape)
    rng.shuffle(pts, axis=0)
    return to_xyzi(pts.astype(np.float32))


def transform_cloud(cloud: np.ndarray, T: np.ndarray) -> np.ndarray:
    xyz = xyz_of(cloud).astype(np.float64)
    out = (xyz @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    return to_xyzi(out, intensity=cloud["intensity"] if cloud.dtype == POINT_XYZI_DTYPE else None)


def pose_error(T_a: np.ndarray, T_b: np.ndarray):
    """(translation error [m], rotation error [rad]) between two 4x4 poses."""
    D = np.linalg.inv(np.asarray(T_a, dtype=np.float64)) @ np.asarray(T_b, dtype=np.float64)
    # rotation angle from the skew part (well conditioned near 0, unlike arccos of the trace of float32 matrices)
    R = D[:3, :3]
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    return float(np.linalg.norm(D[:3, 3])), float(np.arctan2(s, c))


def make_dense_pair(scene_seed: int, n_points: int, extent: float = 60.0):
    """Config 5 pair: two independent area-uniform samplings of one scene; the source is expressed in a frame
    displaced by a known relative pose.  No scan-ring structure, so the GICP optimum sits at the ground truth."""
    scene = make_scene(scene_seed)
    rel = random_relative_pose(2000 + scene_seed)
    tgt = dense_surface_cloud(scene, n_points, 4000 + scene_seed, extent=extent)
    src_w = dense_surface_cloud(scene, n_points, 5000 + scene_seed, extent=extent)
    src = transform_cloud(src_w, np.linalg.inv(rel))
    return tgt, src, rel
