"""Prefilter ("next" row f2: apps/prefiltering_nodelet.cpp:131-182): the oracle against an independent numpy
restatement, the ABI, and (-m gpu) the HIP pipeline against the oracle — point for point, in order."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from hdl_graph_slam_amd import synth


def _scan(seed=3, n_extra=40):
    scene = synth.make_scene(seed)
    cloud = synth.scan(scene, "VLP-16", synth.pose_matrix([0, 0, 0], [0, 0, 0]), 100 + seed)
    rng = np.random.default_rng(seed)
    cloud["intensity"] = rng.uniform(0, 255, len(cloud)).astype(np.float32)
    # a few isolated far-away points (outliers) and two non-finite records
    extra = synth.to_xyzi(rng.uniform(-80, 80, (n_extra, 3)).astype(np.float32) * [1, 1, 0.2], rng.uniform(0, 255, n_extra))
    bad = synth.to_xyzi(np.array([[np.nan, 0, 0], [1, np.inf, 2]], np.float32))
    return np.concatenate([cloud, extra, bad])


def _np_distance_voxel(cloud, near, far, leaf):
    xyz = synth.xyz_of(cloud).astype(np.float32)
    inten = cloud["intensity"].astype(np.float32)
    d = np.sqrt((xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]) + xyz[:, 2] * xyz[:, 2]).astype(np.float64)
    keep = (d > near) & (d < far)
    xyz, inten = xyz[keep], inten[keep]
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(xyz * inv).astype(np.int64)
    mn = np.floor(xyz.min(axis=0) * inv).astype(np.int64)
    div = np.floor(xyz.max(axis=0) * inv).astype(np.int64) - mn + 1
    key = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    ends = np.concatenate([starts[1:], [len(ks)]])
    out = np.zeros((len(starts), 4), np.float32)
    for v, (a, b) in enumerate(zip(starts, ends)):
        acc = np.zeros(4, np.float32)
        for i in order[a:b]:
            acc = (acc + np.array([xyz[i, 0], xyz[i, 1], xyz[i, 2], inten[i]], np.float32)).astype(np.float32)
        out[v] = acc / np.float32(b - a)
    return out


def test_oracle_distance_filter_and_voxelgrid_match_numpy():
    cloud = _scan()
    p = O.default_prefilter_params()
    p.outlier_removal_method = 0
    p.downsample_resolution = 0.25
    got = O.prefilter(cloud, p)
    ref = _np_distance_voxel(cloud[np.isfinite(synth.xyz_of(cloud)).all(axis=1)], 1.0, 100.0, 0.25)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_oracle_outlier_removal_properties():
    cloud = _scan()
    base = O.default_prefilter_params()
    base.outlier_removal_method = 0
    pts = O.prefilter(cloud, base)
    from scipy.spatial import cKDTree
    tree = cKDTree(pts[:, :3].astype(np.float64))
    # radius: kept <=> at least min_neighbors OTHER points within the radius
    pr = O.default_prefilter_params()
    pr.outlier_removal_method, pr.radius_radius, pr.radius_min_neighbors = 2, 0.5, 2
    kept = O.prefilter(cloud, pr)
    cnt = np.array([len(v) for v in tree.query_ball_point(pts[:, :3].astype(np.float64), 0.5 * (1 - 1e-7))]) - 1
    borderline = np.array([len(v) for v in tree.query_ball_point(pts[:, :3].astype(np.float64), 0.5 * (1 + 1e-6))]) - 1
    sure = cnt == borderline
    expect = pts[(cnt >= 2) | ~sure]
    assert len(kept) <= len(expect) and len(kept) >= int(((cnt >= 2) & sure).sum())
    # statistical: idempotent ordering, removes the isolated points, keeps most of the scan
    ps = O.default_prefilter_params()
    kept_s = O.prefilter(cloud, ps)
    assert 0.7 * len(pts) < len(kept_s) < len(pts)
    d, _ = tree.query(kept_s[:, :3].astype(np.float64), k=1)
    assert d.max() < 1e-6    # a subset of the downsampled points, unchanged


def test_prefilter_abi_defaults():
    from hdl_graph_slam_amd import _lib as L
    assert C.sizeof(L.HgsPrefilterParams) == C.sizeof(O.PrefilterParams) == 64
    p = L.HgsPrefilterParams()
    assert L.lib().hgs_prefilter_params_default(C.byref(p)) == 0
    q = O.default_prefilter_params()
    assert all(getattr(p, n) == getattr(q, n) for n, _ in L.HgsPrefilterParams._fields_)


@pytest.mark.gpu
@pytest.mark.parametrize("outlier", [0, 1, 2])
@pytest.mark.parametrize("leaf", [0.1, 0.5, None])
def test_hip_prefilter_matches_oracle(outlier, leaf):
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    cloud = _scan(5)
    p = L.HgsPrefilterParams()
    L.lib().hgs_prefilter_params_default(C.byref(p))
    p.outlier_removal_method = outlier
    p.radius_radius, p.radius_min_neighbors = 0.5, 2
    if leaf is None:
        p.downsample_method = 0
    else:
        p.downsample_resolution = leaf
    reg = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
    dc = reg.prefilter(cloud, p)
    got = dc.download()
    ref = O.prefilter(cloud, p)
    assert len(got) == len(ref), (len(got), len(ref))
    g4 = np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1)
    assert np.array_equal(g4, ref)      # float centroids accumulated in the same order: bit-identical, same order
    # the resident result is a registration input without a second upload
    reg.setInputTarget(dc)
    reg.setInputSource(dc)
    r = reg.align(np.eye(4))
    assert r.converged and np.abs(r.matrix() - np.eye(4)).max() < 1e-5
    dc.close()
    reg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("leaf,radius,min_neighbors,use_filter", [(0.25, 0.5, 2, 1), (0.25, 0.5, 2, 0), (0.3, 0.8, 5, 1), (0.4, 1.6, 12, 1), (0.2, 0.2, 1, 1), (0.5, 0.25, 0, 1)])
def test_hip_prefilter_on_the_voxel_grid_equals_the_separate_passes(leaf, radius, min_neighbors, use_filter):
    """Round 6: with VoxelGrid in front, the distance filter runs inside the voxel grid's kernels and RadiusOutlierRemoval counts neighbours by voxel key
    (k_pf_grid_radius_flags) instead of on a search tree built for the purpose.  Same cloud, same order, as the separate passes (engine option
    prefilter_fast = 0) and as the oracle — the KITTI launch file's parameters first, then radii of 1 .. 4 voxel widths, radii below one, NaN / far points."""
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    cloud = _scan(7)
    rng = np.random.default_rng(11)
    # stragglers that only the outlier removal drops, points the distance filter drops, and non-finite returns
    extra = synth.to_xyzi(np.concatenate([rng.uniform(-80, 80, (300, 3)) * [1, 1, 0.05] + [0, 0, 20.0], rng.uniform(-0.05, 0.05, (20, 3)),
                                          np.array([[np.nan, 1, 1], [np.inf, 0, 0], [500.0, 0, 0]])]).astype(np.float32))
    cloud = np.concatenate([cloud[:4000], extra, cloud[4000:]])
    p = L.HgsPrefilterParams()
    L.lib().hgs_prefilter_params_default(C.byref(p))
    p.use_distance_filter, p.distance_near_thresh, p.distance_far_thresh = use_filter, 0.1, 100.0
    p.downsample_method, p.downsample_resolution = L.HGS_DOWNSAMPLE_VOXELGRID, leaf
    p.outlier_removal_method, p.radius_radius, p.radius_min_neighbors = L.HGS_OUTLIER_RADIUS, radius, min_neighbors
    out = {}
    for fast in (1, 0):
        reg = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
        reg.set_option("prefilter_fast", fast)
        got = reg.prefilter(cloud, p).download()
        out[fast] = np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1)
        reg.close()
    assert out[1].tobytes() == out[0].tobytes()
    ref = O.prefilter(cloud, p)
    assert len(out[1]) == len(ref) and np.array_equal(out[1], ref, equal_nan=True)
    assert 0 < len(ref) < len(cloud)


@pytest.mark.gpu
def test_hip_prefilter_edge_cases():
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    reg = RegistrationHIP(L.default_params(L.HGS_NDT_OMP))
    empty = reg.prefilter(np.zeros((0, 4), np.float32))
    assert empty.size == 0 and len(empty.download()) == 0
    near = synth.to_xyzi(np.array([[0.1, 0.1, 0.0], [0.2, 0.0, 0.1]], np.float32))   # everything inside distance_near_thresh
    assert reg.prefilter(near).size == 0
    one = synth.to_xyzi(np.array([[5.0, 1.0, 0.5]], np.float32), [7.0])
    p = L.HgsPrefilterParams()
    L.lib().hgs_prefilter_params_default(C.byref(p))
    p.outlier_removal_method = 0
    out = reg.prefilter(one, p).download()
    assert len(out) == 1 and out["intensity"][0] == 7.0 and out["x"][0] == 5.0
    reg.close()


@pytest.mark.gpu
def test_cloud_download_round_trip():
    """hgs_cloud_download of an uploaded cloud returns x, y, z and the PointXYZI intensity unchanged, in input order."""
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    cloud = _scan(7)[:2000]
    reg = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
    got = reg.upload(cloud).download()
    for f in ("x", "y", "z", "intensity"):
        assert np.array_equal(got[f], cloud[f], equal_nan=True)
    reg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,layout", [(70001, "xyzi"), (200000, "xyzi"), (60000, "xyz12"), (60000, "xyz16"), (16384, "xyzi"), (0, "xyzi")])
def test_large_upload_round_trip_through_the_pack_pool(n, layout):
    """Uploads are packed by the host into pinned memory — clouds of >= 49152 points by the engine's helper threads, chunk by chunk, each chunk's DMA
    under the packing of the next (hgs_engine.hip, upload_points_packed): every layout and size class comes back bit for bit, NaN records included,
    and repeated uploads reuse the pinned image safely."""
    from hdl_graph_slam_amd import _lib as L, synth
    from hdl_graph_slam_amd.registration import RegistrationHIP
    rng = np.random.default_rng(n + len(layout))
    xyz = rng.normal(0, 20, (n, 3)).astype(np.float32)
    if n > 10:
        xyz[rng.integers(0, n, 7)] = np.nan
    reg = RegistrationHIP(L.default_params(L.HGS_NDT_OMP))
    for rep in range(3):    # the second and third upload meet the previous one's pinned buffers
        xyz_r = xyz + np.float32(rep)
        if layout == "xyzi":
            cloud = synth.to_xyzi(xyz_r, intensity=rng.random(n).astype(np.float32))
        elif layout == "xyz12":
            cloud = np.ascontiguousarray(xyz_r)
        else:
            cloud = np.zeros((n, 4), np.float32)
            cloud[:, :3] = xyz_r
        d = reg.upload(cloud)
        got = d.download()
        assert d.size == n
        want = synth.xyz_of(cloud) if n else np.zeros((0, 3), np.float32)
        for k, f in enumerate(("x", "y", "z")):
            assert np.array_equal(got[f], want[:, k], equal_nan=True), (layout, rep, f)
        if layout == "xyzi":
            assert np.array_equal(got["intensity"], cloud["intensity"])
        else:
            assert not got["intensity"].any()
        d.close()
    if n > 0:
        # ... and the way down: hgs_transform_source (align()'s output cloud) of the same size class — one copy + a scatter spread over the same
        # helper threads for large clouds, four overlapped pieces for small ones — against the float arithmetic of the device's transform
        reg.setInputSource(cloud)
        T = synth.pose_matrix([0.5, -0.25, 0.125], [0.0, 0.0, 0.0])   # a pure translation by exactly representable amounts: no rounding to argue about
        out = reg.transformed_source(T)
        want = synth.xyz_of(cloud) + np.array([0.5, -0.25, 0.125], np.float32)
        assert np.array_equal(out[:, :3], want, equal_nan=True) and (out[:, 3] == 1.0).all()
    reg.close()


def _py_approx_voxelgrid(cloud, leaf):
    """pcl::ApproximateVoxelGrid as PCL runs it — a plain sequential loop over the points with the 512-entry history table —
    independent of oracle/prefilter.hpp and of the device's sort-based form."""
    inv = np.float32(1.0) / np.float32(leaf)
    hist = {}
    out = []

    def flush(e):
        out.append((e[4] / np.float32(e[3])).astype(np.float32))

    for r in cloud:
        p = np.array([r["x"], r["y"], r["z"], r["intensity"]], np.float32)
        if not np.isfinite(p[:3]).all():
            continue
        ix, iy, iz = (int(np.floor(np.float32(p[k] * inv))) for k in range(3))
        h = (ix * 7171 + iy * 3079 + iz * 4231) & 511
        e = hist.get(h)
        if e is not None and e[3] and (e[0], e[1], e[2]) != (ix, iy, iz):
            flush(e)
            e = None
        if e is None:
            e = [ix, iy, iz, 0, np.zeros(4, np.float32)]
            hist[h] = e
        e[3] += 1
        e[4] = (e[4] + p).astype(np.float32)
    for h in sorted(hist):
        if hist[h][3]:
            flush(hist[h])
    return np.array(out, np.float32).reshape(-1, 4)


@pytest.mark.parametrize("leaf", [0.1, 0.5, 2.0])
def test_oracle_approx_voxelgrid_matches_the_sequential_filter(leaf):
    cloud = _scan(4)[::3]
    p = O.default_prefilter_params()
    p.use_distance_filter, p.outlier_removal_method, p.downsample_method, p.downsample_resolution = 0, 0, 2, leaf
    got = O.prefilter(cloud, p)
    ref = _py_approx_voxelgrid(cloud, leaf)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    # the approximate filter emits a voxel once per eviction: never fewer outputs than pcl::VoxelGrid
    p.downsample_method = 1
    assert len(got) >= len(O.prefilter(cloud, p))


def _check_approx_voxelgrid(make_engine):
    from hdl_graph_slam_amd import _lib as L
    reg = make_engine()
    for seed, leaf, dist in ((5, 0.1, 0), (6, 0.5, 1), (7, 3.0, 0)):
        cloud = _scan(seed)
        p = L.HgsPrefilterParams()
        L.lib().hgs_prefilter_params_default(C.byref(p))
        p.use_distance_filter, p.outlier_removal_method = dist, 0
        p.downsample_method, p.downsample_resolution = L.HGS_DOWNSAMPLE_APPROX_VOXELGRID, leaf
        dc = reg.prefilter(cloud, p)
        got = dc.download()
        ref = O.prefilter(cloud, p)
        g4 = np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1)
        assert g4.shape == ref.shape and np.array_equal(g4, ref), (seed, leaf)   # same centroids in the same (eviction) order
        dc.close()
    reg.close()


@pytest.mark.gpu
def test_hip_approx_voxelgrid_matches_oracle_in_order():
    """Row a3 / f2: APPROX_VOXELGRID (scan_matching_odometry_nodelet.cpp:91-96, prefiltering_nodelet.cpp:59-63) on the device —
    the sort-and-scan form reproduces the sequential filter's output INCLUDING its order and its repeated voxels."""
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    _check_approx_voxelgrid(lambda: RegistrationHIP(L.default_params(L.HGS_FAST_GICP)))


# ---- deskewing (apps/prefiltering_nodelet.cpp:182-243) -----------------------------------------------------------------
def _np_deskew(xyz, imu_w, scan_period):
    """Independent numpy restatement: Eigen's float quaternion arithmetic, point by point."""
    f = np.float32
    w = [-f(imu_w[0]), -f(imu_w[1]), -f(imu_w[2])]
    n = len(xyz)
    out = np.empty_like(xyz)
    with np.errstate(invalid="ignore", over="ignore"):
        for i in range(n):
            dt = np.float64(scan_period) * np.float64(i) / np.float64(n)
            q = [f(dt / 2.0 * np.float64(w[0])), f(dt / 2.0 * np.float64(w[1])), f(dt / 2.0 * np.float64(w[2])), f(1.0)]      # x y z w
            n2 = (q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3])
            iv = np.array([-q[0] / n2, -q[1] / n2, -q[2] / n2], f)
            iw = q[3] / n2
            v = xyz[i].astype(f)
            uv = np.array([iv[1] * v[2] - iv[2] * v[1], iv[2] * v[0] - iv[0] * v[2], iv[0] * v[1] - iv[1] * v[0]], f)
            uv = uv + uv
            c = np.array([iv[1] * uv[2] - iv[2] * uv[1], iv[2] * uv[0] - iv[0] * uv[2], iv[0] * uv[1] - iv[1] * uv[0]], f)
            out[i] = (v + iw * uv) + c
    return out


def test_oracle_deskew_matches_numpy_and_undoes_a_constant_rate_rotation():
    cloud = _scan(4)[:3000]
    p = O.default_prefilter_params()
    p.use_distance_filter, p.downsample_method, p.outlier_removal_method = 0, 0, 0
    imu_w, period = [0.3, -0.2, 1.1], 0.1
    got = O.prefilter(cloud, p, imu_angular_velocity=imu_w, scan_period=period)
    xyz = np.stack([cloud["x"], cloud["y"], cloud["z"]], axis=1)
    ref = _np_deskew(xyz, imu_w, period)
    assert np.array_equal(got[:, :3], ref, equal_nan=True) and np.array_equal(got[:, 3], cloud["intensity"])
    # the meaning: a static point seen by a sensor turning at rate w appears, at time t, rotated by R(-w t) in the sensor frame;
    # deskewing brings every point of the sweep back into the frame of the sweep's start, to first order in |w| t
    rng = np.random.default_rng(0)
    n = 2000
    world = rng.uniform(-20, 20, (n, 3))
    w = np.array(imu_w)
    seen = np.empty((n, 3), np.float32)
    for i in range(n):
        th = w * (period * i / n)
        a = np.linalg.norm(th)
        k = th / a if a > 0 else th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
        seen[i] = R.T @ world[i]
    rec = O.prefilter(synth.to_xyzi(seen), p, imu_angular_velocity=imu_w, scan_period=period)[:, :3]
    assert np.abs(rec - world).max() < 0.02 and np.abs(seen - world).max() > 1.0      # |w| t = 0.12 rad: second-order residue only
    # no gyro sample -> untouched
    assert np.array_equal(O.prefilter(cloud, p)[:, :3], xyz, equal_nan=True)


def test_select_imu_sample_follows_the_nodelet_queue_rule():
    from hdl_graph_slam_amd.registration import select_imu_sample
    assert select_imu_sample([], 5.0) is None
    q = [(1.0, "a"), (2.0, "b"), (3.0, "c"), (4.0, "d")]
    assert select_imu_sample(q, 2.0) == "c" and [s for s, _ in q] == [3.0, 4.0]      # first stamp AFTER the cloud; older ones dropped, it stays
    assert select_imu_sample(q, 2.5) == "c" and len(q) == 2
    assert select_imu_sample(q, 9.0) == "d" and q == []                                 # none newer: the newest, queue emptied
    q = [(7.0, "x")]
    assert select_imu_sample(q, 1.0) == "x" and q == [(7.0, "x")]


def _check_deskewed_prefilter(make_engine):
    from hdl_graph_slam_amd import _lib as L
    reg = make_engine()
    for seed, imu_w, period, full in ((5, [0.3, -0.2, 1.1], 0.1, True), (6, [0.0, 0.0, -2.5], 0.05, False), (7, [0.0, 0.0, 0.0], 0.1, False)):
        cloud = _scan(seed)
        p = L.HgsPrefilterParams()
        L.lib().hgs_prefilter_params_default(C.byref(p))
        if not full:
            p.use_distance_filter, p.downsample_method, p.outlier_removal_method = 0, 0, 0
        got = reg.prefilter(cloud, p, imu_angular_velocity=imu_w, scan_period=period).download()
        ref = O.prefilter(cloud, p, imu_angular_velocity=imu_w, scan_period=period)
        g4 = np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1)
        if not full:   # nothing filtered: the non-finite records are still there (hgs_cloud keeps n_input, downloads them as they are)
            assert g4.shape == ref.shape and np.array_equal(g4, ref, equal_nan=True), seed
        else:
            assert g4.shape == ref.shape and np.array_equal(g4, ref), seed
    # NULL gyro sample = the nodelet's empty queue
    cloud = _scan(8)
    a = reg.prefilter(cloud).download()
    b = reg.prefilter(cloud, None, imu_angular_velocity=None).download()
    assert np.array_equal(a["x"], b["x"])
    reg.close()


@pytest.mark.gpu
def test_hip_deskewed_prefilter_matches_oracle():
    """Row f2: the deskewing step in front of the prefilter, on the device, bit for bit the oracle's (same unfused float sequence)."""
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    _check_deskewed_prefilter(lambda: RegistrationHIP(L.default_params(L.HGS_FAST_GICP)))
