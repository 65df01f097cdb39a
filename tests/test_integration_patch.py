"""The reference-side binding as an artefact: integration/hdl_graph_slam_hip.patch.

CPU (this container, where /root/reference exists):
  * `git apply --check` accepts the patch on the reference tree as it is;
  * the patched src/hdl_graph_slam/registrations.cpp and a translation unit that includes the patched include/hdl_graph_slam/loop_detector.hpp
    compile (with and without -DUSE_HGS_HIP) against stand-in ROS / PCL / Eigen / g2o / ndt_omp / fast_gicp headers (tests/mock_*);
  * linked with the host emulation of the kernels the patched code RUNS: the reference's own select_registration_method returns the HIP
    adapter for the three new names and its poses equal the oracle's; the reference's own LoopDetector::detect picks the oracle's candidate
    through the batched path and through the sequential fall-back (which exercises getFitnessScore on PCL's lazily built CPU tree).
GPU (`-m gpu`): the same binary linked with libhgs_hip.so, prebuilt here by __graft_entry__.build(), against the Python mirror on the device."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import integration_build as IB  # noqa: E402

needs_reference = pytest.mark.skipif(not IB.have_reference(), reason="the reference tree (/root/reference) is only present in the build container")


@needs_reference
def test_patch_applies_to_the_reference_tree():
    subprocess.run(["git", "-C", IB.REFERENCE, "apply", "--check", "--whitespace=nowarn", IB.PATCH], check=True)
    touched = set()
    for line in open(IB.PATCH):
        if line.startswith("+++ b/"):
            touched.add(line[6:].strip())
    assert touched == set(IB.PATCHED)
    # everything the patch adds to C++ sources sits behind USE_HGS_HIP: a build without the backend is the reference's build
    depth, guarded = 0, True
    for line in open(IB.PATCH):
        if line.startswith("+++ b/"):
            cpp = not line.strip().endswith("CMakeLists.txt")
            depth = 0
        elif line.startswith("+") and cpp:
            body = line[1:].strip()
            if body.startswith("#ifdef USE_HGS_HIP"):
                depth += 1
            elif body.startswith("#endif") and depth > 0:
                depth -= 1
            elif body and depth == 0:
                guarded = False
    assert guarded


@needs_reference
def test_patched_reference_sources_compile_against_the_stubs():
    out = IB.build("hip")
    assert out and os.path.exists(out)


def _write_pair(tmp_path):
    from hdl_graph_slam_amd import synth
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.3)
    tgt.tofile(tmp_path / "t.bin")
    src.tofile(tmp_path / "s.bin")
    return tgt, src


def _write_loop_set(tmp_path):
    from hdl_graph_slam_amd import workloads, _lib as L
    wl = workloads.make_loop_closure_set("VLP-16", 3, n_candidates=5, n_distinct=3, downsample=0.4)
    wl.target.tofile(tmp_path / "t.bin")
    np.stack([L.colmajor16(g) for g in wl.guesses]).astype(np.float32).tofile(tmp_path / "g.bin")
    files = []
    for i, c in enumerate(wl.candidates):
        c.tofile(tmp_path / f"c{i}.bin")
        files.append(str(tmp_path / f"c{i}.bin"))
    return wl, files


def _pose(line):
    return np.array([float(v) for v in line.split()[1:]], np.float32).reshape(4, 4).T


def _check_factory(out, align_reference):
    """out: stdout lines of `integration_main factory`; align_reference(method_id, resolution) -> (converged, 4x4) of the same pair."""
    from hdl_graph_slam_amd import _lib as L
    lines = [ln for ln in out if ln.startswith(("factory", "pose"))]
    want = {"FAST_GICP_HIP": L.HGS_FAST_GICP, "FAST_VGICP_HIP": L.HGS_FAST_VGICP, "NDT_HIP": L.HGS_NDT_OMP}
    for head, pose in zip(lines[0::2], lines[1::2]):
        f = head.split()
        name, is_hip, converged, builds = f[1], int(f[3]), int(f[5]), int(f[7])
        if name in want:
            assert is_hip == 1 and converged == 1
            assert builds == 0                       # align() through the reference's factory never built PCL's CPU kd-tree
            conv, T = align_reference(want[name])
            assert conv
            yield name, _pose(pose), T
        else:
            assert is_hip == 0 and converged == 0    # NDT_OMP: the reference's own branch (here a stub engine), untouched by the patch


def _check_loop(out, best_want, T_want, tol):
    det = [i for i, ln in enumerate(out) if ln.startswith("detection")]
    assert len(det) == 2
    for i in det:
        assert out[i].split()[3] == "1", out[i]                # one loop found
        assert int(out[i + 1].split()[1]) == best_want, (out[i + 1], best_want)
        T = _pose(out[i + 2])
        assert np.abs(T - T_want).max() < tol, np.abs(T - T_want).max()
    return [int(out[i].split()[5]) for i in det]               # PCL kd-tree builds so far, per detection


@needs_reference
def test_patched_factory_and_loop_detector_run_on_the_emulated_kernels(tmp_path):
    exe = IB.build("simt")
    if exe is None:
        pytest.skip("no clang++ for the host emulation")
    import oracle as O
    from hdl_graph_slam_amd import synth
    tgt, src = _write_pair(tmp_path)
    out = subprocess.run([exe, "factory", str(tmp_path / "t.bin"), str(tmp_path / "s.bin")], check=True, capture_output=True, text=True).stdout.splitlines()

    def oracle_align(method):
        p = O.default_params(method)
        p.resolution = 1.0
        o = O.OracleRegistration(p)
        o.setInputTarget(tgt)
        o.setInputSource(src)
        r = o.align(np.eye(4))
        return bool(r.converged), r.matrix()
    n = 0
    for name, T, T_oracle in _check_factory(out, oracle_align):
        dt, dr = synth.pose_error(T.astype(np.float64), T_oracle)
        assert dt < 1e-4 and dr < 1e-4, (name, dt, dr)
        n += 1
    assert n == 3

    wl, files = _write_loop_set(tmp_path)
    p = O.default_params(O.HGS_FAST_GICP)
    o = O.OracleRegistration(p)
    o.setInputTarget(wl.target)
    best, best_score, T_best = -1, np.finfo(np.float64).max, None
    for i, (c, g) in enumerate(zip(wl.candidates, wl.guesses)):      # loop_detector.hpp:135-154
        o.setInputSource(c)
        r = o.align(g)
        s = o.getFitnessScore(4.0)
        if r.converged and not s > best_score:
            best, best_score, T_best = i, s, r.matrix()
    for devices, builds_want in (("1", [0, 0]), ("4096", None)):     # batched path; sequential fall-back (the matcher cannot get device 1..4095)
        out = subprocess.run([exe, "loop", "FAST_GICP_HIP", devices, str(tmp_path / "t.bin"), str(tmp_path / "g.bin"), *files], check=True, capture_output=True, text=True).stdout.splitlines()
        builds = _check_loop(out, best, T_best, 2e-4)
        if builds_want is not None:
            assert builds == builds_want        # the batched path never touches PCL's CPU tree
        else:
            assert builds == [1, 1]    # the sequential path calls getFitnessScore: ONE lazy build for the five candidates; the second detection has the same target (same pointer)


@pytest.mark.gpu
def test_patched_factory_and_loop_detector_on_the_gpu(tmp_path):
    exe = IB.build("hip")
    assert exe is not None and os.path.exists(exe), "integration/_build/integration_main is built by __graft_entry__.build() where /root/reference exists"
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    from hdl_graph_slam_amd.registrations import select_registration_method
    tgt, src = _write_pair(tmp_path)
    out = subprocess.run([exe, "factory", str(tmp_path / "t.bin"), str(tmp_path / "s.bin")], check=True, capture_output=True, text=True).stdout.splitlines()

    def mirror_align(method):
        p = L.default_params(method)
        p.resolution = 1.0
        reg = RegistrationHIP(p)
        reg.setInputTarget(tgt)
        reg.setInputSource(src)
        r = reg.align(np.eye(4))
        res = bool(r.converged), r.matrix()
        reg.close()
        return res
    n = 0
    for name, T, T_mirror in _check_factory(out, mirror_align):
        assert np.array_equal(T, T_mirror), name     # the same library on the same inputs: identical bits
        n += 1
    assert n == 3

    wl, files = _write_loop_set(tmp_path)
    reg = select_registration_method({"registration_method": "FAST_GICP"})
    reg.setInputTarget(wl.target)
    rec, best = reg.loop_match_batch([reg.upload(c) for c in wl.candidates], wl.guesses, 4.0)
    T_best = np.array(rec["final_transformation"][best]).reshape(4, 4).T
    reg.close()
    for devices in ("1", "4096"):
        out = subprocess.run([exe, "loop", "FAST_GICP_HIP", devices, str(tmp_path / "t.bin"), str(tmp_path / "g.bin"), *files], check=True, capture_output=True, text=True).stdout.splitlines()
        # (LoopDetector rebuilds the guess from the pose-graph estimates through a quaternion: it differs from the mirror's in the last bits)
        _check_loop(out, int(best), T_best, 2e-4)


# ------------------------------------------------------------------------------------------------------------------ round 6: more hunks
def _next_row_inputs(tmp_path):
    """Committed clouds (tests/golden): the VLP-16 pair of the reference-code pins, three keyframes with poses, one raw sweep."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_refpin_golden as MG
    from hdl_graph_slam_amd import synth
    c, z, n = MG.small_clouds()
    for k in ("pair_target", "pair_source", "kf0", "kf1", "kf2"):
        c[k].tofile(tmp_path / (k + ".bin"))
    np.stack([p.T.reshape(-1) for p in n["poses"]]).astype(np.float32).tofile(tmp_path / "poses.bin")
    synth.to_xyzi(n["raw_xyzi"][:, :3], n["raw_xyzi"][:, 3]).tofile(tmp_path / "raw.bin")
    return z, n, np.load(os.path.join(ROOT, "tests", "golden", "refpin_v1.npz"))


def _run(exe, *args):
    return subprocess.run([exe, *[str(a) for a in args]], check=True, capture_output=True, text=True).stdout.splitlines()


def _check_next_rows(exe, tmp_path):
    """f1 / f3 / f2 through the patched reference functions, next to the UNPATCHED translation units linked into the same binary."""
    from hdl_graph_slam_amd import synth
    z, n, gold = _next_row_inputs(tmp_path)
    # f1: InformationMatrixCalculator::calc_fitness_score / calc_information_matrix (information_matrix_calculator.cpp:25-80)
    for case in (0, 1, 3, 4, 15):     # ground truth at DBL_MAX / 4.0 / 0.25 / no inlier, the perturbed pose
        T, r = gold["fit_pose"][case], float(gold["fit_max_range"][case])
        out = _run(exe, "infomat", tmp_path / "pair_target.bin", tmp_path / "pair_source.bin", *[repr(float(v)) for v in T.reshape(-1)], "max" if r > 1e300 else repr(r))
        assert not [ln for ln in out if ln.startswith("error")], out
        dev = [ln.split() for ln in out if ln.startswith("fitness_device")]
        cpu = [ln.split() for ln in out if ln.startswith("fitness_cpu")][0]
        assert [int(d[3]) for d in dev] == [1, 2] and int(dev[1][5]) == 2 and int(dev[1][7]) > 0      # both calls on the device, two resident clouds with a size
        assert float(cpu[1]) == float(gold["fit_score"][case])                                         # the unpatched unit IS the code that made the vectors
        assert int(cpu[3]) >= 1                                                                        # ... and it built PCL's CPU tree; the device path never did before it
        for d in dev:
            s = float(d[1])
            assert s == float(cpu[1]) if float(cpu[1]) > 1e300 else abs(s / float(cpu[1]) - 1) < 2e-6
        inf_dev = np.array([float(v) for v in [ln for ln in out if ln.startswith("infomat_device")][0].split()[1:]])
        inf_cpu = np.array([float(v) for v in [ln for ln in out if ln.startswith("infomat_cpu")][0].split()[1:]])
        assert np.allclose(inf_dev, inf_cpu, rtol=1e-5, atol=0)
    # f3: MapCloudGenerator::generate (map_cloud_generator.cpp:13-51)
    out = _run(exe, "mapcloud", 0.4, tmp_path / "poses.bin", tmp_path / "dev.bin", tmp_path / "cpu.bin", tmp_path / "kf0.bin", tmp_path / "kf1.bin", tmp_path / "kf2.bin")
    head = out[0].split()
    assert head[0] == "mapcloud" and head[2] == head[4] and int(head[6]) == 1 and int(head[8]) == 3, out      # one device call, three resident keyframes
    dev, cpu = np.fromfile(tmp_path / "dev.bin", synth.POINT_XYZI_DTYPE), np.fromfile(tmp_path / "cpu.bin", synth.POINT_XYZI_DTYPE)
    assert len(dev) == len(n["map_cloud"]) and dev.tobytes() == cpu.tobytes()                                 # voxel centres, order, intensity 0, data[3] = 1
    assert np.array_equal(np.stack([dev["x"], dev["y"], dev["z"]], 1), n["map_cloud"][:, :3])
    # f2: the prefilter hunk's parameter mapping + device call (apps/prefiltering_nodelet.cpp:50-99, 131-133)
    out = _run(exe, "prefilter", tmp_path / "raw.bin", tmp_path / "pf.bin", "APPROX_VOXELGRID", 0.3, "NONE")
    assert out[0].split()[2] == "1", out
    pf = np.fromfile(tmp_path / "pf.bin", synth.POINT_XYZI_DTYPE)
    assert np.array_equal(np.stack([pf["x"], pf["y"], pf["z"], pf["intensity"]], 1), n["approx_voxelgrid_0.3"])


def _check_prefilter_nodelet(kind, tmp_path):
    """f2, the nodelet itself: the reference's PrefilteringNodelet::onInit / initialize_params / cloud_callback from the PATCHED apps/prefiltering_nodelet.cpp,
    once through the USE_HGS_HIP hunk (one hgs_prefilter call) and once with use_hip_prefilter=false (its own filter chain): the published clouds are identical,
    header included — for the KITTI launch file's parameters, the nodelet's defaults (VOXELGRID 0.1 + STATISTICAL) and APPROX_VOXELGRID + RADIUS."""
    from hdl_graph_slam_amd import synth
    exe = IB.exe(kind, "prefilter_nodelet_main")
    assert os.path.exists(exe)
    _z, n, _gold = _next_row_inputs(tmp_path)
    for params in (["downsample_method=VOXELGRID", "downsample_resolution=0.25", "outlier_removal_method=RADIUS", "radius_radius=0.5", "radius_min_neighbors=2", "distance_near_thresh=0.1"],
                   [],
                   ["downsample_method=APPROX_VOXELGRID", "downsample_resolution=0.3", "outlier_removal_method=RADIUS", "radius_radius=1.0", "radius_min_neighbors=2"]):
        out = _run(exe, tmp_path / "raw.bin", tmp_path / "nd.bin", tmp_path / "nc.bin", *params)
        line = [ln for ln in out if ln.startswith("prefilter_nodelet")][0].split()
        assert not [ln for ln in out if ln.startswith("error")], out
        n_dev, n_cpu, calls, calls_after = int(line[4]), int(line[6]), int(line[8]), int(line[10])
        assert n_dev == n_cpu > 0 and calls == 1 and calls_after == 1, line       # the hunk took the device path once; use_hip_prefilter=false did not
        assert open(tmp_path / "nd.bin", "rb").read() == open(tmp_path / "nc.bin", "rb").read()
    got = np.fromfile(tmp_path / "nd.bin", synth.POINT_XYZI_DTYPE)
    assert np.array_equal(np.stack([got["x"], got["y"], got["z"], got["intensity"]], 1), _approx_radius_golden(n))


def _approx_radius_golden(n):
    """the last parameter set above through the oracle, on the committed raw sweep"""
    import oracle as O
    from hdl_graph_slam_amd import synth
    p = O.default_prefilter_params()
    p.downsample_method, p.downsample_resolution, p.outlier_removal_method, p.radius_radius, p.radius_min_neighbors = 2, 0.3, 2, 1.0, 2
    return O.prefilter(synth.to_xyzi(n["raw_xyzi"][:, :3], n["raw_xyzi"][:, 3]), p)


def _check_odometry_nodelet(kind, tmp_path, make_registration, tol):
    """a2 / a4 with the reference's own caller: the PATCHED apps/scan_matching_odometry_nodelet.cpp (its class lives in the .cpp) + the patched factory run a
    6-sweep stream — onInit, cloud_callback, matching (:165-262), publish_odometry and, with a status subscriber, publish_scan_matching_status (:298-335) through the
    USE_HGS_HIP hunk — against hdl_graph_slam_amd.odometry.ScanMatchingOdometry (the Python restatement of the same function) over `make_registration(pnh)`: the
    same odometry, converged flags, matching error (getFitnessScore) and inlier fraction; PCL's CPU kd-tree is never built."""
    from hdl_graph_slam_amd import synth, workloads
    from hdl_graph_slam_amd.odometry import ScanMatchingOdometry
    exe = IB.exe(kind, "odometry_nodelet_main")
    assert os.path.exists(exe)
    st = workloads.make_odometry_stream("VLP-16", 0, 6, speed=2.0, downsample=0.3)
    files = []
    for i, c in enumerate(st.scans):
        c.tofile(tmp_path / f"sw{i}.bin")
        files.append(tmp_path / f"sw{i}.bin")
    for method, mirror_method in (("FAST_GICP_HIP", "FAST_GICP"), ("NDT_HIP", "NDT_OMP")):
        # run A: keyframe switches on (0.5 m), no status subscriber; run B: one keyframe, status subscribed (the status is published BEFORE a keyframe switch,
        # so the mirror's scores are only comparable while the target stays)
        for delta_trans, delta_time, subscribers in ((0.5, 1.0, 0), (5.0, 100.0, 1)):
            out = _run(exe, method, subscribers, "downsample_method=NONE", f"keyframe_delta_trans={delta_trans}", f"keyframe_delta_time={delta_time}", "reg_resolution=1.0", "--", *files)
            odom = [ln.split() for ln in out if ln.startswith("odom")]
            status = {int(ln.split()[1]): ln.split() for ln in out if ln.startswith("status")}
            assert len(odom) == 6 and all(o[-1] == "0" for o in odom), out          # every sweep published; no CPU kd-tree build anywhere (status path included)
            assert len(status) == (5 if subscribers else 0)
            reg = make_registration({"registration_method": mirror_method, "reg_resolution": 1.0})
            od = ScanMatchingOdometry(reg, keyframe_delta_trans=delta_trans, keyframe_delta_angle=0.15, keyframe_delta_time=delta_time)     # (:76-78)
            for k, c in enumerate(st.scans):
                T = od.matching(100.0 + 0.1 * k, c)
                p = np.array([float(v) for v in odom[k][2:5]])
                assert np.abs(p - T[:3, 3]).max() < tol, (method, k, p, T[:3, 3])
                qw, qx, qy, qz = [float(v) for v in odom[k][5:9]]
                R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                              [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                              [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
                assert np.abs(R - T[:3, :3]).max() < 10 * tol, (method, k)
                if k == 0 or not subscribers:
                    continue
                s = status[k]
                assert int(s[3]) == int(od.last_result.converged) == 1
                fit = reg.getFitnessScore()
                assert abs(float(s[5]) - fit) <= 1e-5 * max(fit, 1e-3) + 10 * tol, (method, k, s[5], fit)
                aligned = synth.transform_cloud(c, np.asarray(od.last_result.matrix(), np.float64))
                _idx, d2 = reg.nn_target(synth.xyz_of(aligned))
                frac = float(np.mean(np.asarray(d2).reshape(-1) < 0.25))
                assert abs(float(s[7]) - frac) < 2e-3, (method, k, s[7], frac)
            if subscribers:
                assert od.num_keyframes == 1
            elif mirror_method == "FAST_GICP":
                assert od.num_keyframes >= 2          # run A really switched keyframes (setInputTarget on the new one)
            if hasattr(reg, "close"):
                reg.close()


def _check_regularization_param(exe, tmp_path, align_with):
    """reg_regularization_method reaches both GICP branches of the patched factory (registrations.cpp:27-56) and changes the result like the mirror's."""
    from hdl_graph_slam_amd import _lib as L
    tgt, src = _write_pair(tmp_path)
    poses = {}
    for name, value in (("PLANE", L.HGS_REG_PLANE), ("FROBENIUS", L.HGS_REG_FROBENIUS), ("NOT_A_METHOD", L.HGS_REG_FROBENIUS)):
        out = _run(exe, "factory", tmp_path / "t.bin", tmp_path / "s.bin", name)
        lines = [ln for ln in out if ln.startswith(("factory", "pose"))]
        for head, pose in zip(lines[0::2], lines[1::2]):
            f = head.split()
            if f[1] in ("FAST_GICP_HIP", "FAST_VGICP_HIP"):
                assert int(f[f.index("regularization") + 1]) == value, head
                poses[(f[1], name)] = _pose(pose)
            elif f[1] == "NDT_HIP":
                assert int(f[3]) == 1
    for engine in ("FAST_GICP_HIP", "FAST_VGICP_HIP"):
        assert not np.array_equal(poses[(engine, "PLANE")], poses[(engine, "FROBENIUS")])
        assert np.array_equal(poses[(engine, "NOT_A_METHOD")], poses[(engine, "FROBENIUS")])       # unknown name: warning + the default
    if align_with is not None:
        for engine, method in (("FAST_GICP_HIP", "FAST_GICP"), ("FAST_VGICP_HIP", "FAST_VGICP")):
            for name in ("PLANE", "FROBENIUS"):
                assert np.array_equal(poses[(engine, name)], align_with(method, name, tgt, src)), (engine, name)     # the mirror's pose bit for bit


def _check_loop_robustness(exe, tmp_path, best_want, T_want):
    wl, files = _write_loop_set(tmp_path)
    # a device batch that is refused (the same keyframe twice) must end in the reference's sequential loop, not in a silently dropped detection
    out = _run(exe, "loop_dup", "FAST_GICP_HIP", 1, tmp_path / "t.bin", tmp_path / "g.bin", *files)
    builds = _check_loop(out, best_want, T_want, 2e-4)
    assert builds[0] >= 1, builds             # getFitnessScore on PCL's CPU tree = the sequential loop ran
    # two resident keyframes allowed for five candidates: evicted keyframes are uploaded again, results unchanged
    out = _run(exe, "loop_lru", "FAST_GICP_HIP", 1, tmp_path / "t.bin", tmp_path / "g.bin", *files)
    assert _check_loop(out, best_want, T_want, 2e-4) == [0, 0]


@needs_reference
def test_next_row_hunks_and_robustness_on_the_emulated_kernels(tmp_path):
    exe = IB.build("simt")
    if exe is None:
        pytest.skip("no clang++ for the host emulation")
    import oracle as O
    _check_next_rows(exe, tmp_path)
    _check_prefilter_nodelet("simt", tmp_path)

    def oracle_registration(pnh):
        p = O.default_params(O.HGS_FAST_GICP if pnh["registration_method"] == "FAST_GICP" else O.HGS_NDT_OMP)
        p.resolution = pnh["reg_resolution"]
        return O.OracleRegistration(p)
    _check_odometry_nodelet("simt", tmp_path, oracle_registration, 2e-4)
    _check_regularization_param(exe, tmp_path, None)
    wl, files = _write_loop_set(tmp_path)
    o = O.OracleRegistration(O.default_params(O.HGS_FAST_GICP))
    o.setInputTarget(wl.target)
    best, best_score, T_best = -1, np.finfo(np.float64).max, None
    for i, (c, g) in enumerate(zip(wl.candidates, wl.guesses)):
        o.setInputSource(c)
        r = o.align(g)
        s = o.getFitnessScore(4.0)
        if r.converged and not s > best_score:
            best, best_score, T_best = i, s, r.matrix()
    _check_loop_robustness(exe, tmp_path, best, T_best)


@pytest.mark.gpu
def test_next_row_hunks_and_robustness_on_the_gpu(tmp_path):
    exe = IB.build("hip")
    assert exe is not None and os.path.exists(exe)
    from hdl_graph_slam_amd.registrations import select_registration_method
    _check_next_rows(exe, tmp_path)
    _check_prefilter_nodelet("hip", tmp_path)
    _check_odometry_nodelet("hip", tmp_path, select_registration_method, 2e-6)

    def mirror(method, regularization, tgt, src):
        reg = select_registration_method({"registration_method": method, "reg_regularization_method": regularization, "reg_resolution": 1.0})
        reg.setInputTarget(tgt)
        reg.setInputSource(src)
        T = reg.align(np.eye(4)).matrix()
        reg.close()
        return T
    _check_regularization_param(exe, tmp_path, mirror)
    wl, files = _write_loop_set(tmp_path)
    reg = select_registration_method({"registration_method": "FAST_GICP"})
    reg.setInputTarget(wl.target)
    rec, best = reg.loop_match_batch([reg.upload(c) for c in wl.candidates], wl.guesses, 4.0)
    T_best = np.array(rec["final_transformation"][best]).reshape(4, 4).T
    reg.close()
    _check_loop_robustness(exe, tmp_path, int(best), T_best)
