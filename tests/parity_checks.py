"""Shared parity assertions: an engine under test (the HIP path through the C-ABI on a GPU, or the host execution of
the same device functions in tests/emul) against the CPU oracle on identical inputs.

Tolerances (north_star): final SE(3) pose within 1e-3 m / 1e-3 rad of the oracle; index / integer work bit-exact."""
from __future__ import annotations

import numpy as np

import oracle as O
from hdl_graph_slam_amd import synth

POSE_TOL_M = 1e-3
POSE_TOL_RAD = 1e-3


def make_oracle(params_like) -> O.OracleRegistration:
    p = O.HgsParams()
    for name, _ in O.HgsParams._fields_:
        setattr(p, name, getattr(params_like, name))
    return O.OracleRegistration(p)


def load_pair(engine, oracle, tgt, src):
    for r in (engine, oracle):
        r.setInputTarget(tgt)
        r.setInputSource(src)


def check_nn(engine, oracle, queries):
    """Exact nearest neighbour: indices and float distances bit-identical to the oracle's kd-tree."""
    ie, de = engine.nn_target(queries)
    io, do = oracle.nn_target(queries)
    assert np.array_equal(ie, io), f"{(ie != io).sum()} / {len(io)} nearest-neighbour indices differ"
    assert np.array_equal(de, do)


def check_covariances(engine, tgt, k=20, method=O.HGS_REG_FROBENIUS):
    got = engine.target_covariances(len(tgt)).astype(np.float64)
    ref = O.covariances(tgt, k, method)
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-300)
    rel = (np.abs(got - ref) / scale).max(axis=1)
    if method in (O.HGS_REG_PLANE, O.HGS_REG_MIN_EIG, O.HGS_REG_NORMALIZED_MIN_EIG):
        # U diag(values) V^T is discontinuous where two singular values that receive different values coincide (PLANE on an
        # exactly collinear neighbourhood: the two zero eigen-directions are arbitrary): such points are not comparable
        raw = O.covariances(tgt, k, O.HGS_REG_NONE)
        m = np.zeros((len(raw), 3, 3))
        m[:, 0, 0], m[:, 0, 1], m[:, 0, 2], m[:, 1, 1], m[:, 1, 2], m[:, 2, 2] = raw.T
        m[:, 1, 0], m[:, 2, 0], m[:, 2, 1] = m[:, 0, 1], m[:, 0, 2], m[:, 1, 2]
        w = np.linalg.eigvalsh(m)
        gap = (w[:, 1] - w[:, 0]) if method == O.HGS_REG_PLANE else np.minimum(w[:, 1] - w[:, 0], w[:, 2] - w[:, 1])
        well = gap > 1e-7 * np.maximum(w[:, 2], 1e-300)
        if method != O.HGS_REG_PLANE:
            well |= w[:, 1] < 1e-3 * (w[:, 2] if method == O.HGS_REG_NORMALIZED_MIN_EIG else 1.0)   # both clamped to 1e-3: equal values
        assert well.mean() > 0.9
        rel = rel[well]
    # float storage (6e-8) + different summation order
    assert rel.max() < 5e-6, f"covariance mismatch: max rel {rel.max():.3e}"


def outlier_cloud(seed=5, n_dense=3000, n_far=40):
    """A dense blob plus a few scattered far points: the k nearest neighbours of a far point are spread over the whole blob, so
    the wave that holds it visits hundreds of leaves in the radius pass — more than k_knn_cov's leaf log holds (the gather pass
    then walks the tree instead of replaying the log), next to waves that stay well inside it."""
    rng = np.random.default_rng(seed)
    blob = rng.normal(0, 0.4, (n_dense, 3))
    far = rng.uniform(-60, 60, (n_far, 3)) * [1, 1, 0.1]
    pts = np.concatenate([blob, far]).astype(np.float32)
    return synth.to_xyzi(pts[rng.permutation(len(pts))])


def check_covariances_both_gathers(make_engine):
    """k_knn_cov's two gather passes — the tree walk and the replay of pass 1's leaf log (the harness option knn_replay forces either, the engine
    otherwise chooses by launch shape) — on a LiDAR scan, a tie-heavy cloud and the outlier cloud whose log overflows."""
    import os
    scene = synth.make_scene(3)
    clouds = [synth.scan(scene, "VLP-16", synth.pose_matrix([0, 0, 0], [0, 0, 0]), 31), tie_heavy_cloud(), outlier_cloud()]
    old = os.environ.get("HGS_ENGINE_OPTIONS")
    try:
        per_mode = {}
        for replay in ("0", "1", "2"):
            os.environ["HGS_ENGINE_OPTIONS"] = "knn_replay=" + replay          # applied by the harness when the engine is created
            for ci, cloud in enumerate(clouds):
                e = make_engine(O.default_params(O.HGS_FAST_GICP))
                e.setInputTarget(cloud)
                check_covariances(e, cloud, 20)
                per_mode[(replay, ci)] = e.target_covariances(len(cloud)).copy()
                e.close()
    finally:
        if old is None:
            os.environ.pop("HGS_ENGINE_OPTIONS", None)
        else:
            os.environ["HGS_ENGINE_OPTIONS"] = old


def check_cov_split_equals_inline(make_engine):
    """Non-FROBENIUS regularisations: the search kernel staging fp64 covariances + k_cov_regularize (default) against the single kernel with the
    eigen-decomposition inline (engine option cov_split=0): the same arithmetic on the same fp64 values, identical bits."""
    import os
    scene = synth.make_scene(3)
    cloud = synth.scan(scene, "VLP-16", synth.pose_matrix([0, 0, 0], [0, 0, 0]), 31)[::3]
    old = os.environ.get("HGS_ENGINE_OPTIONS")
    try:
        for method in (O.HGS_REG_PLANE, O.HGS_REG_MIN_EIG, O.HGS_REG_NONE):
            got = {}
            for split in ("0", "1"):
                os.environ["HGS_ENGINE_OPTIONS"] = "cov_split=" + split
                p = O.default_params(O.HGS_FAST_GICP)
                p.regularization_method = method
                e = make_engine(p)
                e.setInputTarget(cloud)
                got[split] = e.target_covariances(len(cloud)).copy()
                e.close()
            assert np.array_equal(got["0"], got["1"]), method
    finally:
        if old is None:
            os.environ.pop("HGS_ENGINE_OPTIONS", None)
        else:
            os.environ["HGS_ENGINE_OPTIONS"] = old


def check_two_launch_rounds_equal_four_launch_rounds(make_engine, reps=3, light=False):
    """Launches of <= 4 GICP problems run an LM round in TWO launches, the control steps replicated in every block of k_gicp_linearize<true> /
    k_gicp_error<true> and the states alternating between two buffers (engine option fused_rounds, the default), against the four-launch round
    (fused_rounds=0) on the same engine: the same code on the same partials in the same order — poses, iteration counts, LM tries, errors and fitness
    scores are identical bits.  Covers an accepted-at-once loop, the LM rejection path (several error / decide rounds per linearisation), an empty source,
    batches of 1 / 3 / 4 / 5 candidates (5: a batch that qualifies by its total tile count — engine option fused_rounds_max_blocks), the size limit and the packet sizes of the
    linearize kernel.  light: smaller clouds and fewer combinations (the host emulation of the kernels runs this check at ~1e-4 of the device's speed)."""
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.8 if light else 0.3)
    tgt2, src2, T2 = synth.make_pair("HDL-32E", 4, downsample=1.0 if light else 0.4)
    p_lm = O.default_params(O.HGS_FAST_GICP)
    p_lm.max_correspondence_distance = 1.0
    p_lm.transformation_epsilon, p_lm.rotation_epsilon = 1e-5, 1e-6
    p_lm.lm_init_lambda_factor = 1e-12
    hard = T @ synth.pose_matrix([0.8, 0.5, 0.1], [0.01, 0.01, 0.08])
    # a registration that runs out of iterations: the result record handed over in host-mapped memory (engine option early_result) says converged = 0
    p_cut = O.default_params(O.HGS_FAST_GICP)
    p_cut.max_iterations = 2
    e = make_engine(p_cut)
    e.setInputTarget(tgt)
    cut = []
    for fused, early in ((0, 1), (1, 1), (1, 0), (1, 1)):
        e.set_option("fused_rounds", fused)
        e.set_option("early_result", early)
        e.setInputSource(src)
        r = e.align(hard)
        cut.append((bytes(r.final_transformation), r.converged, r.iterations, r.lm_tries, r.error))
    assert all(x == cut[0] for x in cut) and not cut[0][1] and cut[0][2] == 2, [x[1:] for x in cut]
    e.close()
    for params, guess in ((O.default_params(O.HGS_FAST_GICP), np.eye(4)), (p_lm, hard)):
        e = make_engine(params)
        e.setInputTarget(tgt)
        runs = {}
        for k, fused in enumerate((0, 1, 1, 0) + (1,) * reps):
            e.set_option("fused_rounds", fused)
            e.set_option("early_result", int(k % 3 != 2))  # (every third run fetches the record with the result kernel + copy)
            e.setInputSource(src)  # (cold: no correspondences left over from the run before)
            r = e.align(guess)
            fit = e.getFitnessScore()
            runs.setdefault(fused, []).append((bytes(r.final_transformation), r.converged, r.iterations, r.lm_tries, r.error, fit))
        assert all(x == runs[0][0] for x in runs[0] + runs[1]), (runs[0][0][1:], [x[1:] for x in runs[1]])
        assert runs[0][0][2] >= 1
        # an empty source: block 0 runs the control steps on zero partials
        empty = src[:0]
        out = []
        for fused in (0, 1):
            e.set_option("fused_rounds", fused)
            e.setInputSource(empty)
            r = e.align(guess)
            out.append((bytes(r.final_transformation), r.converged, r.iterations, r.lm_tries))
        assert out[0] == out[1]
        # the size limit: a source above it takes the four-launch round under the option
        e.set_option("fused_rounds", 1)
        e.set_option("fused_rounds_below", len(src) - 1)
        e.setInputSource(src)
        r = e.align(guess)
        assert (bytes(r.final_transformation), r.converged, r.iterations, r.lm_tries, r.error, e.getFitnessScore()) == runs[0][0]
        e.set_option("fused_rounds_below", 262144)
        # packet sizes of k_gicp_linearize<true>: 16- / 32-query packets search only, the block's last wave redoes the 64-point wave rows (not a bit may move)
        for q in (16, 32, 0) if light else (16, 32, 64, 0):
            e.set_option("nn_qpw", q)
            e.setInputSource(src)
            r = e.align(guess)
            assert (bytes(r.final_transformation), r.converged, r.iterations, r.lm_tries, r.error, e.getFitnessScore()) == runs[0][0], q
            e.setInputSource(src[: len(src) - 77])  # (a last chunk of 51 points behind a different tile count)
            ra = e.align(guess)
            e.set_option("fused_rounds", 0)
            e.setInputSource(src[: len(src) - 77])
            rb = e.align(guess)
            e.set_option("fused_rounds", 1)
            assert (bytes(ra.final_transformation), ra.iterations, ra.lm_tries, ra.error) == (bytes(rb.final_transformation), rb.iterations, rb.lm_tries, rb.error), q
        # batches: one lane, several lanes, above the limit
        clouds = [e.upload(c) for c in (src, src2, src[::2], src2[::3], src[5::3])]
        guesses = [guess, T2, np.eye(4), T2, guess]
        for n in (3, 5) if light else (1, 3, 4, 5):
            recs = []
            for fused, max_blocks in ((0, 100000), (1, 100000), (1, 0)) if light else ((0, 100000), (1, 100000), (1, 100000), (1, 0)):  # (0: batches above 4 problems in four launches)
                e.set_option("fused_rounds", fused)
                e.set_option("fused_rounds_max_blocks", max_blocks)
                rec, best = e.loop_match_batch(clouds[:n], [np.asarray(g, np.float32) for g in guesses[:n]])
                recs.append((rec.tobytes(), best))
            assert all(x == recs[0] for x in recs[1:]), n
        e.set_option("fused_rounds_max_blocks", 640)
        for c in clouds:
            c.close()
        e.close()


def check_covariances_with_short_packets(make_engine):
    """k_knn_cov's pre-fill window: clouds whose LAST packet holds 1 .. 31 points (it borrows the leaves in front of it so that its lists start full), clouds
    smaller than a window, and launches with 16- / 8-query packets (engine option knn_qpw_tiny: every packet then borrows), under all three gather passes."""
    import os
    scene = synth.make_scene(4)
    scan = synth.scan(scene, "VLP-16", synth.pose_matrix([0, 0, 0], [0, 0, 0]), 17)
    rng = np.random.default_rng(11)
    old = os.environ.get("HGS_ENGINE_OPTIONS")
    try:
        for tiny, replay, sizes in ((0, 2, (21, 31, 33, 39, 40, 63, 65, 97, 1029, 2051)), (16, 2, (33, 47, 1029)), (8, 2, (41, 1029)), (0, 1, (33, 1029)), (16, 1, (47,)),
                                    (0, 0, (33, 1029)), (8, 0, (41,))):
            os.environ["HGS_ENGINE_OPTIONS"] = f"knn_qpw_tiny={tiny},knn_replay={replay}"
            for n in sizes:
                for cloud in (scan[:n], synth.to_xyzi(rng.uniform(-3, 3, (n, 3)).astype(np.float32))):
                    e = make_engine(O.default_params(O.HGS_FAST_GICP))
                    e.setInputTarget(cloud)
                    check_covariances(e, cloud, 20)
                    e.close()
    finally:
        if old is None:
            os.environ.pop("HGS_ENGINE_OPTIONS", None)
        else:
            os.environ["HGS_ENGINE_OPTIONS"] = old


def check_covariances_with_outliers(make_engine):
    for k in (20, 48):
        p = O.default_params(O.HGS_FAST_GICP)
        p.correspondence_randomness = k
        e = make_engine(p)
        cloud = outlier_cloud()
        e.setInputTarget(cloud)
        check_covariances(e, cloud, k)
        e.close()


def check_gicp_linearize(engine, oracle, T, err_rel=1e-6):
    He, be, ee, ce = engine.gicp_linearize(T)
    Ho, bo, eo, co = oracle.gicp_linearize(T)
    assert np.array_equal(ce, co), f"{(ce != co).sum()} correspondences differ"
    assert np.abs(He - Ho).max() <= 2e-5 * np.abs(Ho).max()
    assert np.abs(be - bo).max() <= 2e-5 * max(np.abs(bo).max(), 1e-9 * np.abs(Ho).max())
    assert abs(ee - eo) <= err_rel * abs(eo)


def check_align(engine, oracle, guess, tol_m=POSE_TOL_M, tol_rad=POSE_TOL_RAD, same_iterations=True):
    re = engine.align(guess)
    ro = oracle.align(guess)
    dt, dr = synth.pose_error(re.matrix(), ro.matrix())
    assert dt <= tol_m and dr <= tol_rad, f"pose differs from the oracle by {dt:.3e} m / {dr:.3e} rad"
    assert bool(re.converged) == bool(ro.converged)
    if same_iterations:
        assert re.iterations == ro.iterations and re.lm_tries == ro.lm_tries, (re.iterations, ro.iterations, re.lm_tries, ro.lm_tries)
    return re, ro


def check_fitness(engine, oracle, T, max_ranges=(np.finfo(np.float64).max, 4.0, 0.25, 1e-12)):
    for mr in max_ranges:
        fe = engine.getFitnessScore(mr, T=T)
        fo = oracle.getFitnessScore(mr, T=T)
        assert engine.last_num_inliers == oracle.last_num_inliers, (mr, engine.last_num_inliers, oracle.last_num_inliers)
        if oracle.last_num_inliers == 0:
            assert fe == fo == np.finfo(np.float64).max
        else:
            assert abs(fe - fo) <= 1e-9 * abs(fo), (mr, fe, fo)


def check_ndt_cells(engine, oracle):
    ie, me, ce, ne = engine.ndt_cells()
    io, mo, co, no = oracle.ndt_cells()
    assert len(ie) == len(io), (len(ie), len(io))
    key = {tuple(k): i for i, k in enumerate(io)}
    idx = np.array([key[tuple(k)] for k in ie])
    assert len(set(idx.tolist())) == len(idx)
    assert np.array_equal(ne, no[idx])
    assert np.abs(me - mo[idx]).max() < 1e-9
    # inverse covariances are stored in float; the double pipeline behind them is contraction-free on both sides
    assert (np.abs(ce - co[idx]).max(axis=1) / np.abs(co[idx]).max(axis=1)).max() < 1.5e-7


def check_ndt_derivatives(engine, oracle, p6, rel=2e-5):
    se, ge, He = engine.ndt_derivatives(p6)
    so, go, Ho = oracle.ndt_derivatives(p6)
    assert abs(se - so) <= rel * abs(so)
    assert np.abs(ge - go).max() <= rel * np.abs(go).max()
    assert np.abs(He - Ho).max() <= rel * np.abs(Ho).max()


def ndt_guesses(T, n=12, seed=0):
    """Initial guesses for an NDT run: the identity (what the odometry nodelet passes for the first frame after a keyframe,
    scan_matching_odometry_nodelet.cpp:210), the ground truth, and n - 2 perturbations of it (up to 0.3 m / 0.02 rad)."""
    rng = np.random.default_rng(seed)
    out = [np.eye(4), np.array(T, np.float64)]
    for _ in range(n - 2):
        dt = rng.uniform(-0.3, 0.3, 3) * [1, 1, 0.2]
        dr = rng.uniform(-0.02, 0.02, 3) * [0.3, 0.3, 1]
        out.append(T @ synth.pose_matrix(dt, dr))
    return out


def check_ndt_to_convergence(engine, params, tgt, src, guesses, bitwise=True, report=None, label=""):
    """NDT end to end, every run to convergence (no iteration cap below the reference's, no prefix comparison).

    The engine under test adds the per-point contributions of a derivative pass with an order-independent exact
    accumulation; the oracle implements the same definition (sum mode 1) and ndt_omp's serial double sum (sum mode 0).
      * against sum mode 1: the final transformation is BIT-IDENTICAL with equal iteration and derivative-pass counts
        (bitwise=False, KDTREE neighbourhoods whose per-point cell order differs: 1e-9);
      * against sum mode 0 (the upstream-faithful association: ndt_omp adds the per-point results serially in index order, whatever
        its thread count): within the north-star tolerance 1e-3 m / 1e-3 rad with equal iteration counts.  Runs that separate
        are counted and reported; the bound is the measured one (profiles/r03_ndt_sum_mode_deviation.md: 0 of 2100 runs to
        convergence, CPU and GPU together): at most 2 % of the guesses, i.e. none of a set of fewer than 50."""
    exact, serial = make_oracle(params).set_ndt_sum_mode(1), make_oracle(params)
    for o in (exact, serial):
        o.setInputTarget(tgt)
        o.setInputSource(src)
    engine.setInputTarget(tgt)
    engine.setInputSource(src)
    separated, rows = 0, []
    for g in guesses:
        re, rx, rs = engine.align(g), exact.align(g), serial.align(g)
        assert re.iterations == rx.iterations and re.lm_tries == rx.lm_tries and bool(re.converged) == bool(rx.converged), (
            label, re.iterations, rx.iterations, re.lm_tries, rx.lm_tries)
        if bitwise:
            assert bytes(re.final_transformation) == bytes(rx.final_transformation), (label, synth.pose_error(re.matrix(), rx.matrix()))
        else:
            dt, dr = synth.pose_error(re.matrix(), rx.matrix())
            assert dt <= 1e-9 and dr <= 1e-9, (label, dt, dr)
        dt, dr = synth.pose_error(re.matrix(), rs.matrix())
        ok = dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and re.iterations == rs.iterations
        separated += 0 if ok else 1
        rows.append({"iterations": int(re.iterations), "passes": int(re.lm_tries), "converged": int(re.converged), "dt_vs_serial_m": float(dt),
                     "dr_vs_serial_rad": float(dr), "iterations_serial": int(rs.iterations)})
    if report is not None:
        report.append({"case": label, "points": int(len(src)), "guesses": len(guesses), "separated_from_serial_sum": separated,
                       "max_iterations_run": max(r["iterations"] for r in rows), "runs": rows})
    assert separated * 50 <= len(guesses), (label, separated, rows)
    return rows


def write_report(name, payload):
    """Leaves a JSON summary under gpurun_out/ (merged back from the GPU box) next to the test log."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def tie_heavy_cloud(seed: int = 0) -> np.ndarray:
    """Exactly equidistant neighbours at every rank of the k-NN: a regular 0.25 m grid plane (for k = 20 the k-th distance
    is shared by 8 points of which 7 belong to the 20 nearest), the same plane once more (every point duplicated), a
    3-fold duplicated lattice block, vertical collinear runs, plus ordinary random points; shuffled."""
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(np.arange(-4, 4, 0.25, dtype=np.float32), np.arange(-4, 4, 0.25, dtype=np.float32)), -1).reshape(-1, 2)
    single = np.concatenate([g + np.float32([20.0, 0.0]), np.full((len(g), 1), 0.5, np.float32)], 1)
    double = np.concatenate([g, np.full((len(g), 1), 0.5, np.float32)], 1)
    lat = np.stack(np.meshgrid(*[np.arange(0, 2, 0.5, dtype=np.float32)] * 3), -1).reshape(-1, 3) + np.float32([-20.0, 5.0, 0.0])
    t = np.linspace(0, 1, 40, dtype=np.float32)[:, None]
    lines = np.concatenate([np.float32([x0, y0, -1.0]) + t * np.float32([0.0, 0.0, 2.5]) for x0, y0 in rng.uniform(-15, 15, (20, 2)).astype(np.float32)])
    rnd = rng.normal(0, 6, (3000, 3)).astype(np.float32)
    out = np.concatenate([single, double, double, lat, lat, lat, lines.astype(np.float32), rnd])
    return synth.to_xyzi(out[rng.permutation(len(out))])


def check_ndt_edge_cases(make_engine):
    """NDT inputs the nodelets can produce at the borders: an empty source, a source that meets no target cell at all, a target
    without a single valid cell (fewer than 6 points per voxel everywhere), non-finite points, and a one-point source.  The
    engine must terminate, agree with the oracle's exact-sum mode on convergence flag, iteration count and pose, and survive."""
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.4)
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution = 1.0
    far = src.copy()
    far["x"] += 500.0                                   # outside the target's voxel grid: no point meets a cell
    sparse_tgt = tgt[::60]                              # ~70 points over 100 m: no voxel reaches 6 points
    with_nan = src.copy()
    with_nan["x"][::7] = np.nan
    with_nan["z"][3::11] = np.inf
    cases = [("empty source", tgt, src[:0], np.eye(4)), ("source outside the grid", tgt, far, np.eye(4)), ("target without valid cells", sparse_tgt, src, T),
             ("non-finite source points", tgt, with_nan, T), ("one-point source", tgt, src[100:101], T), ("non-finite target points", with_nan, src, np.eye(4))]
    e = make_engine(p)
    for name, t, s, guess in cases:
        o = make_oracle(p).set_ndt_sum_mode(1)
        for r in (e, o):
            r.setInputTarget(t)
            r.setInputSource(s)
        re, ro = e.align(guess), o.align(guess)
        assert bool(re.converged) == bool(ro.converged) and re.iterations == ro.iterations, (name, re.converged, ro.converged, re.iterations, ro.iterations)
        a, b = re.matrix(), ro.matrix()
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        if not np.isnan(b).any():
            assert bytes(re.final_transformation) == bytes(ro.final_transformation), (name, synth.pose_error(a, b))
    e.close()


NDT_DEVIATION_MODES = (("DIRECT7 @ 1.0 m (every launch file)", 1.0, O.HGS_DIRECT7), ("DIRECT1 @ 0.5 m", 0.5, O.HGS_DIRECT1), ("KDTREE @ 1.0 m", 1.0, O.HGS_KDTREE))


def ndt_serial_sum_deviation(make_exact, tgt, src, T, n_guesses=100, seed=0, modes=NDT_DEVIATION_MODES):
    """How far the order-independent exact sum of the per-point NDT contributions (what the device computes; the oracle's sum
    mode 1) ends from the upstream-faithful SERIAL double sum (ndt_omp adds the per-point results in index order: sum mode 0),
    run for run, to convergence.  make_exact(params) -> an engine with setInputTarget / setInputSource / align (the HIP engine, or
    the oracle in sum mode 1).  Returns one record per neighbourhood mode: the number of guesses that end more than the north-star
    tolerance (1e-3 m / 1e-3 rad) apart or with another iteration count, and the largest differences."""
    out = []
    guesses = ndt_guesses(T, n=n_guesses, seed=seed)
    for label, res, search in modes:
        p = O.default_params(O.HGS_NDT_OMP)
        p.resolution, p.neighbor_search = res, search
        exact, serial = make_exact(p), make_oracle(p)
        for r in (exact, serial):
            r.setInputTarget(tgt)
            r.setInputSource(src)
        beyond, other_iters, dts, drs = 0, 0, [], []
        for g in guesses:
            re, rs = exact.align(g), serial.align(g)
            dt, dr = synth.pose_error(re.matrix(), rs.matrix())
            dts.append(float(dt)), drs.append(float(dr))
            beyond += 0 if (dt <= POSE_TOL_M and dr <= POSE_TOL_RAD) else 1
            other_iters += 0 if re.iterations == rs.iterations else 1
        out.append({"mode": label, "resolution": res, "neighbor_search": int(search), "guesses": len(guesses), "beyond_1e-3": beyond,
                    "other_iteration_count": other_iters, "max_dt_m": max(dts), "max_dr_rad": max(drs),
                    "p99_dt_m": float(np.percentile(dts, 99)), "median_dt_m": float(np.median(dts))})
        if hasattr(exact, "close"):
            exact.close()
    return out


def check_nn1_with_equidistant_targets(make_engine):
    """The 1-NN of k_gicp_linearize / k_fitness where several target points are EXACTLY equally near: the quad walk keeps only the
    minimum distance per leaf and resolves the index from the lane's best leaf, and must fall back to the exact keyed walk when the
    minimum was reached in two different leaves.  Target: the tie-heavy cloud (duplicated planes, a 3-fold lattice, a regular grid);
    source: its own points at the identity (distance 0 to every copy of a duplicated point: the lowest original index wins) and
    shifted by exactly half a grid step (0.125 m: every grid query is equidistant to two grid points — float-exact arithmetic)."""
    tgt = tie_heavy_cloud()
    src = tgt[::3].copy()
    p = O.default_params(O.HGS_FAST_GICP)
    e, o = make_engine(p), make_oracle(p)
    load_pair(e, o, tgt, src)
    for shift in ([0.0, 0.0, 0.0], [0.125, 0.0, 0.0], [0.125, 0.125, 0.0], [0.0, 0.0, 0.25]):
        T = synth.pose_matrix(shift, [0.0, 0.0, 0.0])
        check_gicp_linearize(e, o, T)
        check_gicp_linearize(e, o, T)      # again, now seeded by the correspondences just found (the unordered walk)
        check_fitness(e, o, T, max_ranges=(np.finfo(np.float64).max, 0.02))
    if hasattr(e, "close"):
        e.close()


def check_nn1_on_small_trees(make_engine, sizes=(21, 24, 25, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 511, 513, 1025)):
    """The quad walk on every small tree shape: 3 to 129 leaves, odd and even heights (an odd height starts at the virtual node 0),
    partly filled last leaves, the top quad that reaches past a tiny tree's own groups — correspondences (bit-exact) and fitness
    against the oracle, unseeded and seeded, for sources larger and smaller than a packet."""
    rng = np.random.default_rng(11)
    p = O.default_params(O.HGS_FAST_GICP)
    for n in sizes:
        tgt = synth.to_xyzi(rng.normal(0, 2.0, (n, 3)).astype(np.float32))
        for m in (7, 150):
            src = synth.to_xyzi((rng.normal(0, 2.0, (m, 3)) + [0.3, 0.0, 0.0]).astype(np.float32))
            if m < 21:
                src = np.concatenate([src, synth.to_xyzi(rng.normal(0, 2.0, (21 - m, 3)).astype(np.float32))])   # covariances need k + 1 points
            e, o = make_engine(p), make_oracle(p)
            load_pair(e, o, tgt, src)
            T = synth.pose_matrix([0.05, -0.02, 0.01], [0.0, 0.0, 0.01])
            check_gicp_linearize(e, o, T)
            check_gicp_linearize(e, o, synth.pose_matrix([0.06, -0.02, 0.01], [0.0, 0.0, 0.012]))   # seeded by the first call
            check_fitness(e, o, T, max_ranges=(np.finfo(np.float64).max, 1.0))
            if hasattr(e, "close"):
                e.close()
    # one or two leaves (no boxes to test: the walk scans the leaves) and a single point: getFitnessScore needs no covariances
    pn = O.default_params(O.HGS_NDT_OMP)
    for n in (1, 3, 8, 9, 16):
        tgt = synth.to_xyzi(rng.normal(0, 2.0, (n, 3)).astype(np.float32))
        src = synth.to_xyzi(rng.normal(0, 2.0, (70, 3)).astype(np.float32))
        e, o = make_engine(pn), make_oracle(pn)
        load_pair(e, o, tgt, src)
        check_fitness(e, o, synth.pose_matrix([0.1, 0.0, 0.0], [0.0, 0.0, 0.02]), max_ranges=(np.finfo(np.float64).max, 4.0))
        if hasattr(e, "close"):
            e.close()


def check_adapter_lazy_tree_lines(out):
    """Lines 6-9 of tests/cpp/adapter_main.cpp: pcl::Registration's CPU kd-tree behind the adapter (adapters/registration_hip.hpp, LazyKdTree).
    align() -> initCompute() builds nothing; the non-virtual getFitnessScore() / getSearchMethodTarget() through the base pointer build it once and
    agree with the device; a new target does not build again; setAlignedCloudOutput(false) leaves align()'s copy of the input in `output`."""
    assert out[6] == "cpu_tree_builds_after_aligns 0 built 0", out[6]
    f = out[7].split()
    assert f[0] == "cpu_tree_builds_after_getFitnessScore" and int(f[1]) == 1
    fit_hip = float(out[2].split()[1])
    assert abs(float(f[3]) - fit_hip) <= 1e-6 * max(1.0, abs(fit_hip)), (f[3], fit_hip)   # both exact 1-NN; the float transform differs by rounding only
    g = out[8].split()
    assert g[0] == "nn_cpu" and g[3] == "nn_hip" and int(g[7]) == 1
    assert int(g[1]) == int(g[4]) and abs(float(g[2]) - float(g[5])) <= 1e-6 * max(1e-6, float(g[2]))
    f = out[9].split()   # the aligned cloud computed on the host (pcl::transformPointCloud's unfused float arithmetic) vs the device's fma chain: float rounding apart
    assert f[0] == "aligned_cloud" and int(f[3]) > 0 and float(f[5]) < 2e-5, out[9]
    assert out[10] == "new_target converged 1 builds 1 output_is_input_copy 1", out[10]
