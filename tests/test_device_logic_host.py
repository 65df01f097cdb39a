"""CPU-side validation of the device code: the HGS_HD per-item functions of hdl_graph_slam_amd/csrc/*.h (tree traversal,
covariance, GICP / NDT per-point terms, LM / Newton state machines) executed on the host by tests/emul and compared with
the oracle.  The kernels proper are compared on a real MI355X in test_hip_parity.py (-m gpu)."""
import numpy as np
import pytest

import oracle as O
import parity_checks as PC
from hdl_graph_slam_amd import synth

emul = pytest.importorskip("emul.emul", reason="host harness needs g++")


def _pair(kind):
    if kind == "vlp16":
        return synth.make_pair("VLP-16", 2, downsample=0.1)
    if kind == "hdl32":
        return synth.make_pair("HDL-32E", 4, downsample=0.25)
    if kind == "dense":
        return synth.make_dense_pair(3, 20000, extent=25.0)
    raise KeyError(kind)


@pytest.fixture(scope="module", params=["vlp16", "hdl32", "dense"])
def gicp_case(request):
    tgt, src, T = _pair(request.param)
    p = O.default_params(O.HGS_FAST_GICP)
    e, o = emul.EmulRegistration(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    return e, o, tgt, src, T


def test_tree_search_is_exact(gicp_case):
    e, o, tgt, src, T = gicp_case
    q = synth.xyz_of(src)[::3]
    PC.check_nn(e, o, q)
    # far-away / degenerate queries exercise the unbounded search and the root-level pruning
    far = np.array([[1e4, 0, 0], [0, -5e3, 30], [0, 0, 0], [1e-3, 1e-3, 1e-3]], np.float32)
    PC.check_nn(e, o, far)
    order = e.sorted_order(True)
    assert np.array_equal(np.sort(order), np.arange(len(tgt)))


def test_covariances(gicp_case):
    e, o, tgt, src, T = gicp_case
    PC.check_covariances(e, tgt)


def test_covariances_with_equidistant_neighbours():
    """Ties at the k-th distance go to the lowest original indices, however many of them there are."""
    cloud = PC.tie_heavy_cloud()
    for k in (20, 10):
        p = O.default_params(O.HGS_FAST_GICP)
        p.correspondence_randomness = k
        e = emul.EmulRegistration(p)
        e.setInputTarget(cloud)
        PC.check_covariances(e, cloud, k)


@pytest.mark.parametrize("method", [O.HGS_REG_PLANE, O.HGS_REG_MIN_EIG, O.HGS_REG_NORMALIZED_MIN_EIG, O.HGS_REG_NONE])
def test_covariance_regularization_methods(method):
    """fast_gicp::RegularizationMethod other than the FROBENIUS default (setRegularizationMethod)."""
    tgt, src, T = _pair("vlp16")
    p = O.default_params(O.HGS_FAST_GICP)
    p.regularization_method = method
    e, o = emul.EmulRegistration(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    PC.check_covariances(e, tgt, 20, method)
    if method != O.HGS_REG_NONE:   # raw covariances of coplanar neighbourhoods are singular: not a usable GICP metric
        # covariances with a 1e-3 : 1 spectrum stored as floats: the Mahalanobis sum carries their 6e-8 rounding amplified
        PC.check_gicp_linearize(e, o, T.astype(np.float32).astype(np.float64), err_rel=2e-5)
        PC.check_align(e, o, T @ synth.pose_matrix([0.2, -0.1, 0.02], [0.002, -0.003, 0.015]), tol_m=1e-6, tol_rad=2e-5)


def test_gicp_linearize(gicp_case):
    e, o, tgt, src, T = gicp_case
    PC.check_gicp_linearize(e, o, T.astype(np.float32).astype(np.float64))
    PC.check_gicp_linearize(e, o, np.eye(4))


@pytest.mark.parametrize("guess_kind", ["identity", "near"])
def test_gicp_align(gicp_case, guess_kind):
    e, o, tgt, src, T = gicp_case
    guess = np.eye(4) if guess_kind == "identity" else T @ synth.pose_matrix([0.2, -0.1, 0.02], [0.002, -0.003, 0.015])
    re, ro = PC.check_align(e, o, guess, tol_m=1e-6, tol_rad=2e-5)
    PC.check_fitness(e, o, ro.matrix())


def test_gicp_lm_rejection_path(gicp_case):
    """Tiny initial lambda factor and a far guess force rejected LM tries (rho < 0) — the retry branch must follow the oracle."""
    e, o, tgt, src, T = gicp_case
    p = O.default_params(O.HGS_FAST_GICP)
    p.max_correspondence_distance = 1.0
    p.transformation_epsilon, p.rotation_epsilon = 1e-5, 1e-6
    e2, o2 = emul.EmulRegistration(p), O.OracleRegistration(p)
    PC.load_pair(e2, o2, tgt, src)
    guess = T @ synth.pose_matrix([0.8, 0.5, 0.1], [0.01, 0.01, 0.08])
    re, ro = PC.check_align(e2, o2, guess, tol_m=1e-5, tol_rad=2e-5)
    assert ro.lm_tries >= ro.iterations


@pytest.fixture(scope="module", params=[("hdl32", 1.0, O.HGS_DIRECT1), ("vlp16", 0.5, O.HGS_DIRECT7), ("dense", 1.0, O.HGS_DIRECT27)])
def vgicp_case(request):
    kind, res, search = request.param
    tgt, src, T = _pair(kind)
    p = O.default_params(O.HGS_FAST_VGICP)
    p.resolution, p.neighbor_search = res, search
    e, o = emul.EmulRegistration(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    return e, o, tgt, src, T


def test_vgicp_linearize(vgicp_case):
    """Voxel lookups (number of voxel correspondences per source point) bit-exact; H, b, error to float-covariance accuracy."""
    e, o, tgt, src, T = vgicp_case
    PC.check_gicp_linearize(e, o, T.astype(np.float32).astype(np.float64))
    PC.check_gicp_linearize(e, o, np.eye(4))


@pytest.mark.parametrize("guess_kind", ["identity", "near"])
def test_vgicp_align(vgicp_case, guess_kind):
    e, o, tgt, src, T = vgicp_case
    guess = np.eye(4) if guess_kind == "identity" else T @ synth.pose_matrix([0.2, -0.1, 0.02], [0.002, -0.003, 0.015])
    PC.check_align(e, o, guess, tol_m=1e-5, tol_rad=2e-5)


@pytest.fixture(scope="module", params=[("hdl32", 1.0, O.HGS_DIRECT7), ("hdl32", 0.5, O.HGS_DIRECT1), ("vlp16", 1.0, O.HGS_DIRECT7),
                                        ("hdl32", 1.0, O.HGS_KDTREE)])
def ndt_case(request):
    kind, res, search = request.param
    tgt, src, T = _pair(kind)
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution, p.neighbor_search = res, search
    e, o = emul.EmulRegistration(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    return e, o, tgt, src, T, kind


@pytest.mark.parametrize("kind", ["vlp16", "hdl32"])
def test_ndt_line_search_follows_the_oracle(kind):
    """hgs_params.ndt_line_search = 1: the More-Thuente state (interval, trial values, stop rule) lives in the device state
    machine, one trial per derivative pass; same iterations, same number of passes, same pose as the oracle's loop."""
    tgt, src, T = _pair(kind)
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution, p.ndt_line_search = 1.0, 1
    e, o = emul.EmulRegistration(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    for off in ([0.0, 0.0, 0.0, 0.0], [0.3, 0.1, 0.0, 0.02], [0.1, -0.05, 0.0, 0.01], [0.02, 0.01, 0.0, 0.002]):
        re, ro = PC.check_align(e, o, T @ synth.pose_matrix(off[:3], [0, 0, off[3]]), tol_m=1e-6, tol_rad=1e-6)
        assert ro.lm_tries >= ro.iterations + 1


def test_ndt_cells(ndt_case):
    e, o, *_ = ndt_case
    PC.check_ndt_cells(e, o)


def test_ndt_derivatives(ndt_case):
    e, o, tgt, src, T, kind = ndt_case
    for p6 in ([T[0, 3], T[1, 3], T[2, 3], 0.003, -0.004, 0.02], [T[0, 3] + 0.1, T[1, 3], T[2, 3], 3.14, 3.13, 3.1], [0, 0, 0, 0, 0, 0]):
        PC.check_ndt_derivatives(e, o, np.array(p6, np.float64), rel=1e-12)


def test_ndt_align_follows_oracle_iteration_by_iteration(ndt_case):
    """ndt_omp's Newton iteration (step clamped to [eps/2, 0.1], no effective line search) is not contractive for every
    guess on weakly constrained scans: a last-bit difference of a re-associated double sum can be amplified until the
    trajectories separate after a dozen iterations (see test_hip_parity.test_ndt_align).  Parity is therefore asserted
    (a) per pass (above), (b) after a fixed, truncated number of iterations, (c) end-to-end from well-behaved guesses."""
    e, o, tgt, src, T, kind = ndt_case
    wild = T @ synth.pose_matrix([0.05, 0.02, 0.0], [0.0, 0.0, 0.004])
    for max_it in (0, 3):
        p = O.default_params(O.HGS_NDT_OMP)
        p.resolution, p.neighbor_search, p.max_iterations = e.params.resolution, e.params.neighbor_search, max_it
        e2, o2 = emul.EmulRegistration(p), O.OracleRegistration(p)
        PC.load_pair(e2, o2, tgt, src)
        re, ro = PC.check_align(e2, o2, wild, tol_m=1e-5, tol_rad=2e-5)
        assert ro.iterations <= max_it + 2
    if kind == "hdl32" and e.params.resolution == 1.0:
        settled = 0
        for off in ([0.02, 0.01, 0.0, 0.002], [0.1, -0.05, 0.0, 0.01], [0.3, 0.1, 0.0, 0.02], [0.1, 0.0, 0.0, 0.01]):
            guess = T @ synth.pose_matrix(off[:3], [0, 0, off[3]])
            ro = o.align(guess)
            if ro.iterations <= 8:       # the oracle settles quickly: the engine must land on the same pose
                PC.check_align(e, o, guess)
                settled += 1
        assert settled >= 2


def test_empty_and_tiny_inputs():
    p = O.default_params(O.HGS_FAST_GICP)
    e = emul.EmulRegistration(p)
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [np.nan, 0, 0]], np.float32)
    e.setInputTarget(tgt)
    idx, d2 = e.nn_target(np.array([[0.9, 0.1, 0.0]], np.float32))
    assert idx[0] == 1
    e.setInputSource(np.zeros((0, 3), np.float32))
    r = e.align(np.eye(4))
    assert r.iterations >= 1   # an empty source yields H = 0: the LM loop runs and terminates without crashing


@pytest.mark.parametrize("variant", [0, 1])
def test_wave_walk_simulation_is_exact(gicp_case, variant):
    """The wave-cooperative 4-ary walk of hgs_wave_bvh.h, simulated lane by lane on the host (both the re-test-on-pop
    and the in-place-leaves / direct-pop scheduling), returns exactly what the per-lane search returns — bounded,
    unbounded and seeded with the previous correspondences."""
    e, o, tgt, src, T = gicp_case
    emul.lib().emul_set_sim_variant(variant)
    try:
        for bound2, seeded in ((np.float32(2.5 ** 2), False), (np.finfo(np.float32).max, False), (np.float32(2.5 ** 2), True)):
            if seeded:
                e.gicp_linearize(np.eye(4))          # leaves the correspondences at the identity pose behind as seeds
            groups, leaves, waves, mismatches = e.walk_stats(T, bound2, seeded)
            assert mismatches == 0 and waves == (len(src) + 63) // 64 and leaves >= waves
    finally:
        emul.lib().emul_set_sim_variant(0)
