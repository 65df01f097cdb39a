"""Map cloud generation ("next" row f3: src/hdl_graph_slam/map_cloud_generator.cpp:13-51): oracle against numpy, and
(-m gpu) the HIP pipeline over resident keyframes against the oracle."""
import numpy as np
import pytest

import oracle as O
from hdl_graph_slam_amd import synth


def _keyframes(seed=4, n_kf=3, spacing=2.0):
    scene = synth.make_scene(seed)
    rng = np.random.default_rng(seed)
    clouds, poses = [], []
    for k in range(n_kf):
        pose = synth.pose_matrix([spacing * k, 0.3 * k, 0.0], [0.0, 0.0, 0.05 * k])
        c = synth.scan(scene, "VLP-16", pose, 300 + k)[::2]
        c["intensity"] = rng.uniform(0, 255, len(c)).astype(np.float32)
        clouds.append(c)
        poses.append(pose)
    clouds[1] = np.concatenate([clouds[1], synth.to_xyzi(np.array([[np.nan, 1, 2]], np.float32))])
    return clouds, poses


def _np_map(clouds, poses, res):
    pts = []
    for c, T in zip(clouds, poses):
        T = np.asarray(T, np.float32)
        x, y, z = c["x"], c["y"], c["z"]
        pts.append(np.stack([((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3] for r in range(3)] + [c["intensity"]], axis=1).astype(np.float32))
    allp = np.concatenate(pts)
    if res <= 0:
        return allp
    fin = np.isfinite(allp[:, :3]).all(axis=1)
    p0 = allp[fin][0, :3].astype(np.float64) - res / 2
    cells = np.floor((allp[fin, :3].astype(np.float64) - p0) / res).astype(np.int64)
    u = np.unique(cells, axis=0)
    u = u[np.lexsort((u[:, 0], u[:, 1], u[:, 2]))]
    out = np.zeros((len(u), 4), np.float32)
    out[:, :3] = ((u + 0.5) * res + p0).astype(np.float32)
    return out


LARGE = dict(n_kf=3, spacing=400.0)   # 0.8 km of trajectory at the launch files' 0.05 / 0.01 m: 10^10 .. 10^12 lattice cells


def _scan_poses(clouds, poses, spacing):
    """_keyframes() ray-casts at the poses; for the large map the scans are simply placed `spacing` apart."""
    return clouds, [synth.pose_matrix([spacing * k, 0.3 * k, 0.0], [0.0, 0.0, 0.05 * k]) for k in range(len(clouds))]


@pytest.mark.parametrize("res", [0.05, 0.01])
def test_oracle_map_cloud_beyond_31_bit_cell_indices(res):
    clouds, poses = _scan_poses(*_keyframes(), spacing=LARGE["spacing"])
    got = O.map_cloud(clouds, poses, res)
    ref = _np_map(clouds, poses, res)
    assert len(ref) > 1000 and got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True)


@pytest.mark.parametrize("res", [0.0, 0.05, 0.5])
def test_oracle_map_cloud_matches_numpy(res):
    clouds, poses = _keyframes()
    got = O.map_cloud(clouds, poses, res)
    ref = _np_map(clouds, poses, res)
    assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("res", [0.0, 0.05, 0.5])
def test_hip_map_cloud_matches_oracle(res):
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    clouds, poses = _keyframes()
    reg = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
    resident = [reg.upload(c) for c in clouds]       # keyframes live on the device (loop-closure candidates)
    m = reg.map_cloud(resident, poses, res).download()
    ref = O.map_cloud(clouds, poses, res)
    got = np.stack([m["x"], m["y"], m["z"], m["intensity"]], axis=1)
    assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True)
    assert reg.map_cloud([], [], res).size == 0
    if res == 0.05:   # a map whose lattice box has more cells than 31 bits index (the launch files' resolution over 0.8 km)
        clouds, poses = _scan_poses(clouds, poses, spacing=LARGE["spacing"])
        for fine in (0.05, 0.01):
            m = reg.map_cloud(resident, poses, fine).download()
            ref = O.map_cloud(clouds, poses, fine)
            assert np.array_equal(np.stack([m["x"], m["y"], m["z"], m["intensity"]], axis=1), ref, equal_nan=True)
    reg.close()
