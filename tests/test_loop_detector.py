"""LoopDetector mirror (include/hdl_graph_slam/loop_detector.hpp): host logic on CPU, the batched device call on GPU."""
import numpy as np
import pytest

from hdl_graph_slam_amd import synth
from hdl_graph_slam_amd.loop_detector import KeyFrame, LoopDetector, loop_guess, sequential_matching


class _NoRegistration:
    pass


def _kf(x, y, accum, yaw=0.0):
    return KeyFrame(cloud=np.zeros(0), estimate=synth.pose_matrix([x, y, 0.3], [0, 0, yaw]), accum_distance=accum)


def test_find_candidates_gating():
    ld = LoopDetector({"distance_thresh": 5.0, "accum_distance_thresh": 8.0, "min_edge_interval": 5.0}, registration=_NoRegistration())
    kfs = [_kf(0, 0, 0.0), _kf(3, 0, 3.0), _kf(30, 0, 30.0), _kf(4, 1, 60.0), _kf(0, 4.9, 95.0)]
    new = _kf(1, 0, 100.0)
    cands = ld.find_candidates(kfs, new)
    # accumulated distance >= 8 for all; planar distance <= 5: kf0 (1.0), kf1 (2.0), kf3 (3.16), kf4 (5.0001 -> out)
    assert [kfs.index(c) for c in cands] == [0, 1, 3]
    assert ld.find_candidates(kfs, _kf(1, 0, 4.0)) == []          # too close to the last loop edge (accum 4 - 0 < 5)
    ld.last_edge_accum_distance = 97.0
    assert ld.find_candidates(kfs, new) == []
    near = _kf(1, 0, 99.0)
    ld.last_edge_accum_distance = 0.0
    assert kfs[4] not in ld.find_candidates(kfs, near)            # 99 - 95 < accum_distance_thresh


def test_loop_guess_matches_reference_formula():
    rng = np.random.default_rng(0)
    a = synth.pose_matrix(rng.normal(size=3), rng.normal(size=3) * 0.2)
    b = synth.pose_matrix(rng.normal(size=3), rng.normal(size=3) * 0.2)
    a[:3, :3] *= 1.0 + 1e-3                                        # drifted (non-orthonormal) rotation, as after graph optimisation
    g = loop_guess(a, b)
    an = a.copy()
    u, _, vt = np.linalg.svd(a[:3, :3])
    an[:3, :3] = u @ vt
    ref = (np.linalg.inv(an) @ b).astype(np.float32)
    assert g[2, 3] == 0.0
    ref[2, 3] = 0.0
    assert np.allclose(g, ref, atol=2e-3) and np.allclose(g[:3, :3] @ g[:3, :3].T, np.eye(3), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["FAST_GICP", "NDT_OMP"])
def test_batched_matching_equals_the_sequential_loop(method):
    """hgs_loop_match_batch == the reference's sequential align/getFitnessScore loop (loop_detector.hpp:135-154): same
    per-candidate poses and scores, same selected candidate."""
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    wl = workloads.make_loop_closure_set("HDL-32E", 3, n_candidates=6, n_distinct=3, downsample=0.25)
    pnh = {"registration_method": method, "reg_resolution": 1.0, "fitness_score_thresh": 2.5, "distance_thresh": 30.0, "accum_distance_thresh": 1.0,
           "min_edge_interval": 0.0}
    new_kf = KeyFrame(wl.target, np.eye(4), 100.0)
    cands = [KeyFrame(c, Tg.copy(), float(i)) for i, (c, Tg) in enumerate(zip(wl.candidates, wl.T_gt))]
    # perturb the estimates so that the guesses are off like accumulated odometry drift
    for k, g in zip(cands, wl.guesses):
        k.estimate = g.astype(np.float64)
    ld = LoopDetector(pnh)
    loop = ld.matching(cands, new_kf)
    rec = ld.last_records
    reg_seq = select_registration_method(pnh)
    best, best_score, rel = sequential_matching(reg_seq, cands, new_kf)
    assert loop is not None and cands.index(loop.key2) == best
    assert np.array_equal(loop.relative_pose, rel)
    assert rec["fitness_score"][best] == best_score
    # every candidate, not only the winner
    for i, c in enumerate(cands):
        reg_seq.setInputSource(c.cloud)
        r = reg_seq.align(loop_guess(new_kf.estimate, c.estimate))
        assert bytes(r.final_transformation) == rec["final_transformation"][i].tobytes()
        assert r.converged == rec["converged"][i] and r.iterations == rec["iterations"][i]
    # detect(): a second detection right after the accepted loop is gated by min_edge_interval
    ld.distance_from_last_edge_thresh = 5.0
    assert ld.find_candidates(cands, KeyFrame(wl.target, np.eye(4), 101.0)) == []
    reg_seq.close()
    ld.registration.close()


@pytest.mark.gpu
def test_information_matrix_fitness_score_on_device():
    """f1: InformationMatrixCalculator::calc_fitness_score (information_matrix_calculator.cpp:49-80) on resident clouds."""
    import oracle as O
    from hdl_graph_slam_amd.registrations import select_registration_method
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.2)
    reg = select_registration_method({"registration_method": "FAST_GICP"})
    c1, c2 = reg.upload(tgt), reg.upload(src)
    o = O.OracleRegistration(O.default_params(O.HGS_FAST_GICP))
    o.setInputTarget(tgt)
    o.setInputSource(src)
    for mr in (np.finfo(np.float64).max, 2.0):
        assert abs(reg.calc_fitness_score(c1, c2, T.astype(np.float32), mr) - o.getFitnessScore(mr, T=T.astype(np.float32))) < 1e-9
    reg.close()


@pytest.mark.gpu
def test_ndt_batches_of_different_shape_back_to_back():
    """One engine, three NDT batches whose lanes get different work plans (tile prefix sums, queue heads; hgs_engine.hip run_batch) with nothing in between:
    every lane must see ITS batch's plan (round-5 advisor finding: lanes 1.. were ordered behind an event recorded before the plan upload).  Records are
    per problem, hence must equal the single registrations bit for bit whatever ran before."""
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    wl = workloads.make_loop_closure_set("HDL-32E", 5, n_candidates=9, n_distinct=4, downsample=0.3)
    pnh = {"registration_method": "NDT_OMP", "reg_resolution": 1.0}
    reg = select_registration_method(pnh)
    reg.setInputTarget(wl.target)
    # different sizes per candidate -> different tile counts per lane
    clouds = [c[: len(c) - 257 * i] for i, c in enumerate(wl.candidates)]
    dev = [reg.upload(c) for c in clouds]
    single = []
    for c, g in zip(dev, wl.guesses):
        reg.setInputSource(c)
        r = reg.align(g)
        single.append((bytes(r.final_transformation), r.iterations, r.converged))
    for order in ([0, 1, 2, 3, 4, 5, 6, 7, 8], [8, 3, 5, 1, 0], [2, 7, 4, 6, 8, 1, 3], [5, 0], [4, 4 - 1, 8, 0, 2, 6, 7, 5, 1]):
        for _ in range(3):
            rec, _best = reg.loop_match_batch([dev[i] for i in order], [wl.guesses[i] for i in order])
            for k, i in enumerate(order):
                assert rec["final_transformation"][k].tobytes() == single[i][0] and rec["iterations"][k] == single[i][1] and rec["converged"][k] == single[i][2], (order, k)
    reg.close()
