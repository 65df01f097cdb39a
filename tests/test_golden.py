"""Known-answer tests against tests/golden/vlp16_pair_seed1.npz (made by tests/golden/make_golden.py; the reference has no
golden vectors of its own — "parity unpinned", these are the pins this repo creates).
CPU: the oracle still reproduces them.  GPU (-m gpu): the HIP path reproduces them through the C-ABI."""
import os

import numpy as np
import pytest

import oracle as O
from hdl_graph_slam_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vlp16_pair_seed1.npz"))


def _check_engine(make, tol_pose, rel_stage, exact_counts=True):
    e = make(O.default_params(O.HGS_FAST_GICP))
    e.setInputTarget(G["target_xyz"])
    e.setInputSource(G["source_xyz"])
    idx, d2 = e.nn_target(G["nn_queries"])
    assert np.array_equal(idx, G["nn_idx"]) and np.array_equal(d2, G["nn_d2"])
    H, b, err, corr = e.gicp_linearize(np.eye(4))
    assert np.array_equal(corr, G["gicp_corr_identity"])
    assert np.abs(H - G["gicp_H_identity"]).max() <= rel_stage * np.abs(G["gicp_H_identity"]).max()
    assert np.abs(b - G["gicp_b_identity"]).max() <= rel_stage * np.abs(G["gicp_b_identity"]).max()
    assert abs(err - G["gicp_err_identity"]) <= rel_stage * G["gicp_err_identity"]
    r = e.align(np.eye(4))
    dt, dr = synth.pose_error(r.matrix(), G["gicp_final"])
    assert dt <= tol_pose and dr <= tol_pose
    assert r.converged == G["gicp_converged"]
    if exact_counts:
        assert r.iterations == G["gicp_iterations"] and r.lm_tries == G["gicp_lm_tries"]
    for mr, (score, n) in zip(G["fitness_ranges"], G["fitness"]):
        s = e.getFitnessScore(mr, T=G["gicp_final"])
        assert e.last_num_inliers == int(n) and abs(s - score) <= 1e-9 * score
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution = 1.0
    n = make(p)
    n.setInputTarget(G["target_xyz"])
    n.setInputSource(G["source_xyz"])
    ijk, mean, icov, npts = n.ndt_cells()
    assert len(ijk) == len(G["ndt_cells_ijk"])
    key = {tuple(k): i for i, k in enumerate(G["ndt_cells_ijk"])}
    order = np.array([key[tuple(k)] for k in ijk])
    assert np.array_equal(npts, G["ndt_cells_n"][order]) and np.abs(mean - G["ndt_cells_mean"][order]).max() < 1e-9
    s, g, Hn = n.ndt_derivatives(G["ndt_p0"])
    assert abs(s - G["ndt_score_p0"]) <= rel_stage * G["ndt_score_p0"]
    assert np.abs(g - G["ndt_g_p0"]).max() <= rel_stage * np.abs(G["ndt_g_p0"]).max()
    assert np.abs(Hn - G["ndt_H_p0"]).max() <= rel_stage * np.abs(G["ndt_H_p0"]).max()
    rn = n.align(G["ndt_guess"])
    dt, dr = synth.pose_error(rn.matrix(), G["ndt_final"])
    assert dt <= tol_pose and dr <= tol_pose and rn.iterations == G["ndt_iterations"]
    return e, n


def test_oracle_reproduces_golden():
    _check_engine(O.OracleRegistration, tol_pose=1e-7, rel_stage=1e-9)
    cov = O.covariances(G["target_xyz"], 20)
    assert np.abs(cov - G["target_cov"]).max() <= 1e-12 * np.abs(G["target_cov"]).max()
    tr = G["gicp_trace"]
    assert len(tr) == G["gicp_iterations"] and np.all(np.diff(tr[:, 12]) <= 0)


@pytest.mark.gpu
def test_hip_reproduces_golden():
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP

    def make(params):
        p = L.HgsParams()
        for name, _ in L.HgsParams._fields_:
            setattr(p, name, getattr(params, name))
        return RegistrationHIP(p)

    e, n = _check_engine(make, tol_pose=1e-3, rel_stage=2e-5)
    cov = e.target_covariances(len(G["target_xyz"]))
    assert (np.abs(cov - G["target_cov"]).max(axis=1) / np.abs(G["target_cov"]).max(axis=1)).max() < 5e-6
    e.close()
    n.close()


# ---- second fixture: FAST_VGICP + prefilter ------------------------------------------------------------------------
G2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vlp16_vgicp_prefilter_seed2.npz"))


def _raw2():
    a = G2["raw_xyzi"]
    return synth.to_xyzi(a[:, :3], a[:, 3])


def _prefilter_cases():
    p = O.default_prefilter_params()
    yield "prefilter_default", p
    q = O.default_prefilter_params()
    q.downsample_resolution, q.outlier_removal_method, q.radius_radius, q.radius_min_neighbors = 0.25, 2, 0.5, 2
    yield "prefilter_kitti_radius", q


def _check_vgicp(e, tol_pose, rel_stage):
    e.setInputTarget(G2["target_xyz"])
    e.setInputSource(G2["source_xyz"])
    H, b, err, hits = e.gicp_linearize(np.eye(4))
    assert np.array_equal(hits, G2["vgicp_hits_identity"])
    assert np.abs(H - G2["vgicp_H_identity"]).max() <= rel_stage * np.abs(G2["vgicp_H_identity"]).max()
    assert np.abs(b - G2["vgicp_b_identity"]).max() <= rel_stage * np.abs(G2["vgicp_b_identity"]).max()
    assert abs(err - G2["vgicp_err_identity"]) <= rel_stage * G2["vgicp_err_identity"]
    r = e.align(np.eye(4))
    dt, dr = synth.pose_error(r.matrix(), G2["vgicp_final"])
    assert dt <= tol_pose and dr <= tol_pose
    assert r.converged == G2["vgicp_converged"] and r.iterations == G2["vgicp_iterations"] and r.lm_tries == G2["vgicp_lm_tries"]


def test_oracle_reproduces_golden_v2():
    for name, p in _prefilter_cases():
        assert np.array_equal(O.prefilter(_raw2(), p), G2[name])
    _check_vgicp(O.OracleRegistration(O.default_params(O.HGS_FAST_VGICP)), tol_pose=1e-7, rel_stage=1e-9)


@pytest.mark.gpu
def test_hip_reproduces_golden_v2():
    import ctypes as C
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    e = RegistrationHIP(L.default_params(L.HGS_FAST_VGICP))
    for name, p in _prefilter_cases():
        q = L.HgsPrefilterParams()
        for f, _ in L.HgsPrefilterParams._fields_:
            setattr(q, f, getattr(p, f))
        got = e.prefilter(_raw2(), q).download()
        assert np.array_equal(np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1), G2[name])
    _check_vgicp(e, tol_pose=1e-5, rel_stage=2e-5)
    e.close()


# ---- a long NDT run, iteration by iteration (tests/golden/make_golden_ndt_trace.py) ------------------------------------------
GN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vlp16_ndt_long_run.npz"))


def _ndt_params():
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution = 1.0
    return p


def test_oracle_reproduces_the_long_ndt_trace():
    """Every iteration of a long NDT run: parameters, score and step length of the oracle's exact-sum mode equal the
    committed trace (the arithmetic is a fixed sequence of IEEE operations: no libm exp, no order-dependent sums)."""
    o = O.OracleRegistration(_ndt_params()).set_ndt_sum_mode(1)
    o.setInputTarget(GN["target_xyz"])
    o.setInputSource(GN["source_xyz"])
    r = o.align(GN["guess"])
    assert r.iterations == int(GN["iterations"]) >= 20 and r.lm_tries == int(GN["passes"])
    tr = o.trace()
    assert tr.shape == GN["trace"].shape and np.abs(tr - GN["trace"]).max() <= 1e-12 * max(1.0, np.abs(GN["trace"]).max())
    assert np.array_equal(np.array(r.final_transformation, np.float32), GN["final"])
    # the serial (ndt_omp) sum ends within the north-star tolerance of it
    dt, dr = synth.pose_error(GN["final"].reshape(4, 4).T.astype(np.float64), GN["final_serial_sum"].reshape(4, 4).T.astype(np.float64))
    assert dt <= 1e-3 and dr <= 1e-3


def _check_long_ndt_run(make):
    e = make(_ndt_params())
    e.setInputTarget(GN["target_xyz"])
    e.setInputSource(GN["source_xyz"])
    r = e.align(GN["guess"])
    assert r.iterations == int(GN["iterations"]) and r.lm_tries == int(GN["passes"]) and bool(r.converged) == bool(GN["converged"])
    assert np.array_equal(np.array(r.final_transformation, np.float32), GN["final"])   # the same float matrix, bit for bit
    e.close()


@pytest.mark.gpu
def test_hip_reproduces_the_long_ndt_run():
    from hdl_graph_slam_amd.registration import RegistrationHIP
    from hdl_graph_slam_amd import _lib as L

    def make(params):
        p = L.HgsParams()
        for name, _ in L.HgsParams._fields_:
            setattr(p, name, getattr(params, name))
        return RegistrationHIP(p)
    _check_long_ndt_run(make)


# ---- third fixture: deskewing, pcl::ApproximateVoxelGrid, the map cloud in octree order (tests/golden/make_golden_v3.py) -----------
G3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vlp16_next_rows_v3.npz"))


def _v3():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_v3", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_v3.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _xyzi(a):
    return synth.to_xyzi(a[:, :3], a[:, 3])


def test_oracle_reproduces_golden_v3():
    m = _v3()
    raw = _xyzi(G3["raw_xyzi"])
    for name, p, w in m.cases():
        assert np.array_equal(O.prefilter(raw, p, imu_angular_velocity=w, scan_period=m.SCAN_PERIOD), G3[name]), name
    kf = [_xyzi(G3[f"kf{k}_xyzi"]) for k in range(3)]
    assert np.array_equal(O.map_cloud(kf, list(G3["poses"]), m.MAP_RES), G3["map_cloud"])


def _check_golden_v3(e):
    from hdl_graph_slam_amd import _lib as L
    m = _v3()
    raw = _xyzi(G3["raw_xyzi"])
    for name, p, w in m.cases():
        q = L.HgsPrefilterParams()
        for f, _ in L.HgsPrefilterParams._fields_:
            setattr(q, f, getattr(p, f))
        got = e.prefilter(raw, q, imu_angular_velocity=w, scan_period=m.SCAN_PERIOD).download()
        assert np.array_equal(np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1), G3[name]), name
    resident = [e.upload(_xyzi(G3[f"kf{k}_xyzi"])) for k in range(3)]
    got = e.map_cloud(resident, list(G3["poses"]), m.MAP_RES).download()
    assert np.array_equal(np.stack([got["x"], got["y"], got["z"], got["intensity"]], axis=1), G3["map_cloud"])   # same centres, same order


@pytest.mark.gpu
def test_hip_reproduces_golden_v3():
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    e = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
    _check_golden_v3(e)
    e.close()
