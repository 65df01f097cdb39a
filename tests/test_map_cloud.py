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


class _PclOctree:
    """A literal pointer octree the way pcl::octree::OctreePointCloud grows one (PCL 1.10 octree_pointcloud.hpp, [UPSTREAM-KNOWLEDGE]):
    nested dicts child index -> node, re-rooted when the box doubles, walked depth first.  The oracle computes the same thing in
    closed form (keys + what they gained from later doublings, sorted by interleaved bits); this is the independent statement."""
    EPS = float(np.finfo(np.float32).eps)

    def __init__(self, res):
        self.res, self.root, self.depth, self.defined = float(res), {}, 0, False
        self.mn, self.mx = [0.0] * 3, [0.0] * 3

    def _adopt(self, p):
        import math
        res = self.res
        while True:
            lower = [p[a] < self.mn[a] for a in range(3)]
            upper = [p[a] >= self.mx[a] for a in range(3)]
            if not (any(lower) or any(upper) or not self.defined):
                return
            if self.defined:
                child = ((not upper[0]) << 2) | ((not upper[1]) << 1) | (not upper[2])
                self.root = {child: self.root}
                side = float(1 << self.depth) * res
                for a in range(3):
                    if not upper[a]:
                        self.mn[a] -= side
                self.depth += 1
                side = float(1 << self.depth) * res - self.EPS
                for a in range(3):
                    self.mx[a] = self.mn[a] + side
            else:
                for a in range(3):
                    self.mn[a], self.mx[a] = p[a] - res / 2, p[a] + res / 2
                # getKeyBitSize() on the empty tree
                mk = [int(math.ceil((self.mx[a] - self.mn[a] - self.EPS) / res)) for a in range(3)]
                self.depth = max(min(32, int(math.ceil(math.log2(max(max(mk), 2)) - self.EPS))), 0)
                side = float(1 << self.depth) * res
                for a in range(3):
                    over = (side - (self.mx[a] - self.mn[a])) / 2.0
                    if over > self.EPS:
                        self.mn[a] -= over
                        self.mx[a] += over
                self.defined = True

    def add(self, p32):
        p = [float(p32[0]), float(p32[1]), float(p32[2])]
        self._adopt(p)
        key = [int((p[a] - self.mn[a]) / self.res) for a in range(3)]
        node = self.root
        for bit in range(self.depth - 1, -1, -1):
            idx = (((key[0] >> bit) & 1) << 2) | (((key[1] >> bit) & 1) << 1) | ((key[2] >> bit) & 1)
            node = node.setdefault(idx, {})

    def centers(self):
        out = []

        def walk(node, key, level):
            for idx in range(8):
                if idx in node:
                    k = [(key[0] << 1) | ((idx >> 2) & 1), (key[1] << 1) | ((idx >> 1) & 1), (key[2] << 1) | (idx & 1)]
                    if level + 1 == self.depth:
                        out.append([np.float32((k[a] + 0.5) * self.res + self.mn[a]) for a in range(3)])
                    else:
                        walk(node[idx], k, level + 1)
        if self.defined:
            walk(self.root, [0, 0, 0], 0)
        return np.array(out, np.float32).reshape(-1, 3)


def _np_map(clouds, poses, res):
    pts = []
    for c, T in zip(clouds, poses):
        T = np.asarray(T, np.float32)
        x, y, z = c["x"], c["y"], c["z"]
        pts.append(np.stack([((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3] for r in range(3)] + [c["intensity"]], axis=1).astype(np.float32))
    allp = np.concatenate(pts)
    if res <= 0:
        return allp
    tree = _PclOctree(res)
    for q in allp[np.isfinite(allp[:, :3]).all(axis=1), :3]:
        tree.add(q)
    c = tree.centers()
    out = np.zeros((len(c), 4), np.float32)
    out[:, :3] = c
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


def test_oracle_map_cloud_growth_in_every_direction_and_first_point_on_a_boundary():
    """Boxes that double towards every octant in turn (the doubling direction depends on which point comes first), duplicates,
    and the structural facts: the first point sits on the corner shared by the 8 voxels of the initial root, so its own voxel
    centre is half a voxel away on every axis; the output is a set of distinct centres whatever the order."""
    for trial, res, ordered in _growth_cases():
        pts, order = ordered, slice(None)
        cloud = synth.to_xyzi(pts[order])
        got = O.map_cloud([cloud], [np.eye(4)], res)
        ref = _np_map([cloud], [np.eye(4)], res)
        assert got.shape == ref.shape and np.array_equal(got, ref), trial
        assert len(np.unique(got[:, :3], axis=0)) == len(got)
        d = np.abs(got[:, :3].astype(np.float64) - pts[order][0].astype(np.float64))
        assert np.isclose(d.min(axis=0), res / 2, atol=1e-5 * max(1.0, 40 / res)).all()     # nearest centre per axis: half a voxel from the first point
        # every input point lies in a returned voxel
        for q in pts[::37]:
            assert (np.abs(got[:, :3].astype(np.float64) - q.astype(np.float64)).max(axis=1) <= res / 2 + 1e-4).any()


def _growth_cases():
    rng = np.random.default_rng(11)
    for trial in range(6):
        res = [0.5, 0.25, 1.0, 0.1, 2.0, 0.3][trial]
        n = 400
        pts = rng.normal(0, 1, (n, 3)).astype(np.float32) * np.float32(3 + 10 * trial)
        pts[1:9] = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float32) * np.float32(40 + 7 * trial)   # early far points: 8 octants
        pts[20] = pts[10]                                                                       # duplicate
        order = rng.permutation(n) if trial % 2 else np.arange(n)
        yield trial, res, pts[order]


def _check_map_growth(make_engine):
    reg = make_engine()
    for trial, res, pts in _growth_cases():
        halves = [synth.to_xyzi(pts[:150]), synth.to_xyzi(np.concatenate([np.array([[np.nan, 0, 0]], np.float32), pts[150:]]))]
        poses = [np.eye(4), np.eye(4)]
        resident = [reg.upload(c) for c in halves]
        m = reg.map_cloud(resident, poses, res).download()
        ref = O.map_cloud(halves, poses, res)
        got = np.stack([m["x"], m["y"], m["z"], m["intensity"]], axis=1)
        assert got.shape == ref.shape and np.array_equal(got, ref), trial      # same voxels in the same (octree traversal) order
    # nothing finite: an empty map, not an error; one finite point: the single voxel whose corner it sits on
    nan = synth.to_xyzi(np.full((5, 3), np.nan, np.float32))
    assert reg.map_cloud([reg.upload(nan)], [np.eye(4)], 0.5).size == 0 and len(O.map_cloud([nan], [np.eye(4)], 0.5)) == 0
    one = synth.to_xyzi(np.array([[np.nan, 0, 0], [1.0, 2.0, 3.0]], np.float32))
    m = reg.map_cloud([reg.upload(one)], [np.eye(4)], 0.5).download()
    ref = O.map_cloud([one], [np.eye(4)], 0.5)
    assert len(m) == 1 and np.array_equal(np.stack([m["x"], m["y"], m["z"], m["intensity"]], axis=1), ref)
    # deeper than 21 levels: refused, not wrong
    from hdl_graph_slam_amd.registration import HgsError
    far = synth.to_xyzi(np.array([[0, 0, 0], [3.0e5, 0, 0]], np.float32))
    with pytest.raises(HgsError):
        reg.map_cloud([reg.upload(far)], [np.eye(4)], 0.01)
    with pytest.raises(ValueError):
        O.map_cloud([far], [np.eye(4)], 0.01)
    reg.close()


@pytest.mark.gpu
def test_hip_map_cloud_replays_the_octree_growth():
    """Row f3: boxes doubling towards every octant, duplicates, a non-finite point between keyframes — centres AND their order."""
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    _check_map_growth(lambda: RegistrationHIP(L.default_params(L.HGS_FAST_GICP)))


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
