"""The PRODUCT kernels and engine on the CPU: hdl_graph_slam_amd/csrc/hgs_kernels.hip and hgs_engine.hip, unchanged, compiled
for the host against the SIMT emulation shim of tests/emul (GPU threads as fibers, the 64 lanes of a wave meeting in
__ballot / __shfl_down / readlane / wave barriers, blocks one after the other, HIP runtime mapped to malloc / memcpy) and
driven through the same C-ABI and the same Python mirror as on the GPU.  What the `-m gpu` parity tests assert on an MI355X
is asserted here on small clouds without one: packet walks, the two-pass k-NN with its tie rule, tile reductions, the
on-device LM / Newton state machines including the three-lane Jacobi SVD, the batch lanes and the progress mirror of the
host loop, prefilter and map-cloud kernels.  (Performance, and anything that depends on real concurrency, is not.)"""
import os
import numpy as np
import pytest

import oracle as O
import parity_checks as PC
from hdl_graph_slam_amd import synth

simt = pytest.importorskip("emul.simt", reason="needs tests/emul")


@pytest.fixture(scope="module", autouse=True)
def simt_library():
    """Points the package's loader at tests/emul/libhgs_simt.so for the duration of this module (and back afterwards)."""
    path = simt.build()
    if path is None:
        pytest.skip("clang++ not available: the emulation build needs ext_vector_type / elementwise builtins")
    from hdl_graph_slam_amd import _lib as L
    saved = (L.LIB_PATH, L._lib)
    L.LIB_PATH, L._lib = path, None
    yield path
    L.LIB_PATH, L._lib = saved


def _engine(params):
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    p = L.HgsParams()
    for name, _ in L.HgsParams._fields_:
        setattr(p, name, getattr(params, name))
    return RegistrationHIP(p)


def _pair(kind):
    if kind == "vlp16":
        return synth.make_pair("VLP-16", 1, downsample=0.3)     # ~7 k points
    if kind == "hdl32":
        return synth.make_pair("HDL-32E", 4, downsample=0.4)    # ~6 k points
    raise KeyError(kind)


@pytest.fixture(scope="module", params=["vlp16", "hdl32"])
def gicp_case(request):
    tgt, src, T = _pair(request.param)
    p = O.default_params(O.HGS_FAST_GICP)
    e, o = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    yield e, o, tgt, src, T
    e.close()


def test_emulation_library_is_not_the_product_library(simt_library):
    from hdl_graph_slam_amd import _lib as L
    assert simt_library.endswith("libhgs_simt.so") and L.LIB_PATH == simt_library
    assert all(hasattr(L.lib(), name) for name in L.EXPORTS)


def test_tree_search_is_exact(gicp_case):
    e, o, tgt, src, T = gicp_case
    PC.check_nn(e, o, synth.xyz_of(src)[::3])
    PC.check_nn(e, o, np.array([[1e4, 0, 0], [0, -5e3, 30], [0, 0, 0], [1e-3, 1e-3, 1e-3]], np.float32))


def test_covariances(gicp_case):
    e, o, tgt, src, T = gicp_case
    PC.check_covariances(e, tgt)


@pytest.mark.parametrize("k", [20, 10])
def test_covariances_with_equidistant_neighbours(k):
    """k_knn_cov's tie handling, including the tie-only re-walks for lanes that need more than four equidistant points."""
    cloud = PC.tie_heavy_cloud()
    p = O.default_params(O.HGS_FAST_GICP)
    p.correspondence_randomness = k
    e = _engine(p)
    e.setInputTarget(cloud)
    PC.check_covariances(e, cloud, k)
    e.close()


def test_covariances_with_both_gather_passes():
    PC.check_covariances_both_gathers(_engine)


def test_split_regularisation_kernel_equals_the_inline_one():
    PC.check_cov_split_equals_inline(_engine)


def test_covariances_of_short_last_packets_and_short_packet_launches():
    PC.check_covariances_with_short_packets(_engine)


def test_two_launch_lm_rounds_equal_the_four_launch_rounds():
    PC.check_two_launch_rounds_equal_four_launch_rounds(_engine, reps=1, light=True)


def test_covariances_when_the_leaf_log_overflows():
    PC.check_covariances_with_outliers(_engine)


def test_gicp_linearize_align_fitness(gicp_case):
    e, o, tgt, src, T = gicp_case
    PC.check_gicp_linearize(e, o, T.astype(np.float32).astype(np.float64))
    PC.check_gicp_linearize(e, o, np.eye(4))
    re, ro = PC.check_align(e, o, np.eye(4), tol_m=1e-6, tol_rad=2e-5)
    PC.check_fitness(e, o, ro.matrix())
    again = e.align(np.eye(4))
    assert bytes(again.final_transformation) == bytes(re.final_transformation)      # deterministic reductions


def test_gicp_lm_rejection_path(gicp_case):
    e, o, tgt, src, T = gicp_case
    p = O.default_params(O.HGS_FAST_GICP)
    p.max_correspondence_distance = 1.0
    p.transformation_epsilon, p.rotation_epsilon = 1e-5, 1e-6
    p.lm_init_lambda_factor = 1e-12
    e2, o2 = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e2, o2, tgt, src)
    guess = T @ synth.pose_matrix([0.8, 0.5, 0.1], [0.01, 0.01, 0.08])
    re, ro = PC.check_align(e2, o2, guess, tol_m=1e-5, tol_rad=2e-5)
    assert ro.lm_tries >= ro.iterations
    e2.close()


@pytest.mark.parametrize("method", [O.HGS_REG_PLANE, O.HGS_REG_NORMALIZED_MIN_EIG])
def test_covariance_regularization_methods(method):
    tgt, src, T = _pair("vlp16")
    p = O.default_params(O.HGS_FAST_GICP)
    p.regularization_method = method
    e, o = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    PC.check_covariances(e, tgt, 20, method)
    PC.check_align(e, o, T @ synth.pose_matrix([0.2, -0.1, 0.02], [0.002, -0.003, 0.015]), tol_m=1e-5, tol_rad=2e-5)
    e.close()


@pytest.mark.parametrize("res,search", [(1.0, O.HGS_DIRECT1), (0.5, O.HGS_DIRECT7), (2.0, O.HGS_DIRECT27)])
def test_vgicp(res, search):
    tgt, src, T = _pair("hdl32")
    p = O.default_params(O.HGS_FAST_VGICP)
    p.resolution, p.neighbor_search = res, search
    e, o = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    PC.check_gicp_linearize(e, o, T.astype(np.float32).astype(np.float64))
    re, ro = PC.check_align(e, o, T @ synth.pose_matrix([0.2, -0.1, 0.02], [0.002, -0.003, 0.015]), tol_m=1e-5, tol_rad=2e-5)
    PC.check_fitness(e, o, ro.matrix())
    e.close()


@pytest.mark.parametrize("kind,res,search", [("hdl32", 1.0, O.HGS_DIRECT7), ("hdl32", 0.5, O.HGS_DIRECT1), ("vlp16", 1.0, O.HGS_DIRECT7), ("hdl32", 1.0, O.HGS_KDTREE)])
def test_ndt(kind, res, search):
    """Voxel table build (hash insertion, stable per-cell accumulation, eigen floor), the one-launch iteration (packed cell
    terms, exact integer reduction through wave / block / atomics, Newton step by the last block): the derivative sums equal
    the oracle's exact-sum mode bit for bit, and so do whole runs to convergence."""
    tgt, src, T = _pair(kind)
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution, p.neighbor_search = res, search
    e, o = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    PC.check_ndt_cells(e, o)
    p6 = np.array([T[0, 3], T[1, 3], T[2, 3], 0.003, -0.004, 0.02])
    PC.check_ndt_derivatives(e, o, p6)
    if search != O.HGS_KDTREE:
        se, ge, He = e.ndt_derivatives(p6)
        sx, gx, Hx = _exact_derivatives(p, tgt, src, p6)
        assert se == sx and np.array_equal(ge, gx) and np.array_equal(He, Hx)
    PC.check_ndt_to_convergence(e, p, tgt, src, PC.ndt_guesses(T, 4, seed=1), bitwise=search != O.HGS_KDTREE, label=f"emulated {kind}")
    e.close()


def _exact_derivatives(p, tgt, src, p6):
    o = O.OracleRegistration(p).set_ndt_sum_mode(1)
    o.setInputTarget(tgt)
    o.setInputSource(src)
    return o.ndt_derivatives(p6)


def test_ndt_line_search():
    """The More-Thuente switch through the real k_ndt_pass and the host loop's round budget."""
    tgt, src, T = _pair("vlp16")
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution, p.ndt_line_search = 1.0, 1
    e, o = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    for off in ([0.3, 0.1, 0.0, 0.02], [0.02, 0.01, 0.0, 0.002]):
        re, ro = PC.check_align(e, o, T @ synth.pose_matrix(off[:3], [0, 0, off[3]]), tol_m=1e-6, tol_rad=1e-6)
    assert ro.lm_tries > ro.iterations + 1
    e.close()


@pytest.mark.parametrize("method", ["FAST_GICP", "NDT_OMP"])
def test_loop_batch_equals_the_sequential_loop(method):
    """hgs_loop_match_batch on 4 lanes (run_batch / drive_lanes / the progress mirror) against one align + getFitnessScore per
    candidate, bit for bit; 7 candidates so that the lanes are uneven."""
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registration import select_best
    from hdl_graph_slam_amd.registrations import select_registration_method
    wl = workloads.make_loop_closure_set("VLP-16", 3, n_candidates=7, n_distinct=3, downsample=0.4)
    pnh = {"registration_method": method, "reg_resolution": 1.0}
    if method == "NDT_OMP":
        pnh["reg_maximum_iterations"] = 6      # 8 Newton iterations per candidate keep the emulated run short
    reg = select_registration_method(pnh)
    reg.setInputTarget(wl.target)
    clouds = [reg.upload(c) for c in wl.candidates]
    rec, best = reg.loop_match_batch(clouds, wl.guesses, 4.0)
    assert best == select_best(rec)
    for i, c in enumerate(clouds):
        reg.setInputSource(c)
        r = reg.align(wl.guesses[i])
        assert bytes(r.final_transformation) == rec["final_transformation"][i].tobytes()
        assert r.converged == rec["converged"][i] and r.iterations == rec["iterations"][i]
        assert reg.getFitnessScore(4.0) == rec["fitness_score"][i]
    reg.close()


def test_prefilter_and_map_cloud_match_the_oracle():
    from hdl_graph_slam_amd import _lib as L
    import ctypes as C
    tgt, src, T = synth.make_pair("VLP-16", 2)
    reg = _engine(O.default_params(O.HGS_FAST_GICP))
    for outlier in (L.HGS_OUTLIER_RADIUS, L.HGS_OUTLIER_STATISTICAL):
        p = L.HgsPrefilterParams()
        L.lib().hgs_prefilter_params_default(C.byref(p))
        p.downsample_resolution, p.outlier_removal_method = 0.3, outlier
        op = O.PrefilterParams()
        for name, _ in O.PrefilterParams._fields_:
            setattr(op, name, getattr(p, name))
        got = reg.prefilter(tgt, p).download()
        ref = O.prefilter(tgt, op)
        assert len(got) == len(ref)
        assert np.array_equal(synth.xyz_of(got), synth.xyz_of(ref))
    kfs = [reg.upload(synth.voxel_downsample(c, 0.3)) for c in (tgt, src)]
    poses = [np.eye(4), T]
    got = reg.map_cloud(kfs, poses, 0.5).download()
    ref = O.map_cloud([synth.voxel_downsample(c, 0.3) for c in (tgt, src)], poses, 0.5)
    assert np.array_equal(np.sort(synth.xyz_of(got).view("f4,f4,f4"), axis=0), np.sort(synth.xyz_of(ref).view("f4,f4,f4"), axis=0))
    reg.close()


def test_edge_cases():
    from hdl_graph_slam_amd.registration import HgsError
    for field, bad in (("correspondence_randomness", 65), ("correspondence_randomness", 0), ("regularization_method", 5), ("resolution", 0.0),
                       ("method", 7)):
        bad_p = O.default_params(O.HGS_FAST_GICP)
        setattr(bad_p, field, bad)
        with pytest.raises(HgsError):
            _engine(bad_p)
    p = O.default_params(O.HGS_FAST_GICP)
    e = _engine(p)
    with pytest.raises(HgsError):
        e.align(np.eye(4))
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [np.nan, 0, 0], [0, 0, 1], [np.inf, 1, 1]], np.float32)
    e.setInputTarget(tgt)
    idx, d2 = e.nn_target(np.array([[0.9, 0.1, 0.0], [0.1, 0.1, 0.8]], np.float32))
    assert list(idx) == [1, 4]
    e.setInputSource(np.zeros((0, 3), np.float32))
    assert e.align(np.eye(4)).iterations >= 1
    assert e.getFitnessScore() == np.finfo(np.float64).max
    tgt2, src2, T = synth.make_pair("VLP-16", 1, downsample=0.4)
    o = O.OracleRegistration(p)
    PC.load_pair(e, o, tgt2, src2[:7])
    PC.check_align(e, o, T, tol_m=1e-4, tol_rad=1e-4)
    e.close()


@pytest.mark.parametrize("method", [0, 2])
def test_cpp_adapter_end_to_end(simt_library, tmp_path, method):
    """adapters/registration_hip.hpp (the pcl::Registration subclass the reference's factory would construct) driven like
    scan_matching_odometry_nodelet.cpp:166-221 by tests/cpp/adapter_main.cpp, linked against the emulated library: the C++
    boundary returns the bits of the Python mirror and the pose of the oracle."""
    import os
    import subprocess
    from hdl_graph_slam_amd import _lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "adapter_main_simt")
    src_cpp = os.path.join(root, "tests", "cpp", "adapter_main.cpp")
    deps = [src_cpp, os.path.join(root, "adapters", "registration_hip.hpp"), os.path.join(root, "include", "hgs_registration.h"), simt_library,
            os.path.join(root, "tests", "mock_pcl", "pcl", "registration", "registration.h"), os.path.join(root, "tests", "mock_pcl", "pcl", "search", "kdtree.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "tests", "mock_pcl"), "-I", os.path.join(root, "tests", "mock_eigen"), "-I", os.path.join(root, "include"), src_cpp, "-o", exe,
                        "-L", os.path.dirname(simt_library), "-l:libhgs_simt.so", f"-Wl,-rpath,{os.path.dirname(simt_library)}"], check=True)
    tgt, src, T = _pair("vlp16")
    tgt.tofile(tmp_path / "t.bin")
    src.tofile(tmp_path / "s.bin")
    out = subprocess.run([exe, str(method), str(tmp_path / "t.bin"), str(tmp_path / "s.bin")], check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "converged 1"
    Tc = np.array([float(v) for v in out[1].split()], np.float32).reshape(4, 4).T
    p = O.default_params(method)
    if method == O.HGS_NDT_OMP:
        p.resolution, p.neighbor_search = 1.0, O.HGS_DIRECT7
    e, o = _engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    r = e.align(np.eye(4))
    assert np.array_equal(Tc, r.matrix())
    assert abs(float(out[2].split()[1]) - e.getFitnessScore()) < 1e-9
    assert out[4] == "no_device converged 0 guess_kept 1"   # an engine that cannot be created does not throw into the caller
    assert out[5] == "recovered converged 1 same_pose 1"                # the engine created later holds the clouds set while creation failed
    PC.check_adapter_lazy_tree_lines(out)
    if method == 0:
        dt, dr = synth.pose_error(Tc.astype(np.float64), o.align(np.eye(4)).matrix())
        assert dt < 1e-5 and dr < 1e-5
    e.close()


# ---- the `-m gpu` tests of the caller-level modules, run as they are against the emulated library (their inputs are small
# enough).  Each wrapper calls the GPU test function directly; the module fixture above has redirected the loader.
def _gpu_test(module, name):
    import importlib
    return getattr(importlib.import_module(module), name)


@pytest.mark.parametrize("method", ["FAST_GICP", "NDT_OMP"])
def test_gpu_suite_odometry_stream(method):
    _gpu_test("test_odometry", "test_hip_stream_follows_the_oracle_stream")(method)


@pytest.mark.parametrize("filter_name,method_id", [("VOXELGRID", 1), ("APPROX_VOXELGRID", 2)])
def test_gpu_suite_odometry_device_downsample(filter_name, method_id):
    _gpu_test("test_odometry", "test_hip_stream_with_device_voxelgrid_downsample")(filter_name, method_id)


def test_gpu_suite_information_matrix_fitness():
    _gpu_test("test_loop_detector", "test_information_matrix_fitness_score_on_device")()


@pytest.mark.parametrize("leaf,radius,min_neighbors,use_filter", [(0.25, 0.5, 2, 1), (0.4, 1.6, 12, 1), (0.5, 0.25, 0, 0)])
def test_gpu_suite_prefilter_on_the_voxel_grid(leaf, radius, min_neighbors, use_filter):
    _gpu_test("test_prefilter", "test_hip_prefilter_on_the_voxel_grid_equals_the_separate_passes")(leaf, radius, min_neighbors, use_filter)


def test_gpu_suite_prefilter_edges_and_download():
    _gpu_test("test_prefilter", "test_hip_prefilter_edge_cases")()
    _gpu_test("test_prefilter", "test_cloud_download_round_trip")()


@pytest.mark.parametrize("n,layout", [(70001, "xyzi"), (60000, "xyz12"), (16384, "xyzi"), (0, "xyzi")])
def test_gpu_suite_large_upload_through_the_pack_pool(n, layout):
    _gpu_test("test_prefilter", "test_large_upload_round_trip_through_the_pack_pool")(n, layout)


@pytest.mark.parametrize("res", [0.5, 0.05])
def test_gpu_suite_map_cloud(res):
    _gpu_test("test_map_cloud", "test_hip_map_cloud_matches_oracle")(res)      # 0.05 includes the > 2^31-cell map


def test_gpu_suite_keyframe_directory(tmp_path):
    _gpu_test("test_keyframe_io", "test_loaded_keyframes_become_resident_candidates")(tmp_path)


def test_gpu_suite_golden_vectors():
    _gpu_test("test_golden", "test_hip_reproduces_golden")()
    _gpu_test("test_golden", "test_hip_reproduces_golden_v2")()


def test_graft_entry_smoke(capsys):
    """__graft_entry__.smoke() — what the driver runs on the GPU box before the bench — against the emulated library."""
    import importlib
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").smoke()
    out = capsys.readouterr().out
    assert "smoke FAST_GICP" in out and "smoke NDT_OMP" in out


def test_gpu_suite_two_engines_from_two_host_threads(gicp_case):
    """The odometry and the loop-closure engine of one process, driven from different threads (one emulated scheduler each)."""
    _gpu_test("test_hip_parity", "test_two_engines_run_concurrently")(gicp_case)


# (test_distributed.test_multi_device_matcher_equals_single_batch also passes here; left to the plugin run — the C++ matcher above
# already drives three engines from three host threads)


@pytest.mark.parametrize("n_engines", [3])
def test_cpp_loop_matcher_end_to_end(simt_library, tmp_path, n_engines):
    """adapters/loop_match_hip.hpp — the replacement of LoopDetector::matching's loop body, with one engine per GPU driven from
    host threads and resident keyframes reused across detections — against the Python mirror's single batch, bit for bit."""
    import os
    import subprocess
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "loop_match_main_simt")
    src_cpp = os.path.join(root, "tests", "cpp", "loop_match_main.cpp")
    deps = [src_cpp, os.path.join(root, "adapters", "loop_match_hip.hpp"), os.path.join(root, "include", "hgs_registration.h"), simt_library]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(root, "include"), src_cpp, "-o", exe,
                        "-L", os.path.dirname(simt_library), "-l:libhgs_simt.so", f"-Wl,-rpath,{os.path.dirname(simt_library)}"], check=True)
    # the resident set stays within its budget over a long run of detections, and an evicted keyframe gives the same record when it comes back
    cap = subprocess.run([exe, "capacity", "0", "60", "3", "8", "0", "200"], check=True, capture_output=True, text=True).stdout.split()
    cap = {cap[i]: int(cap[i + 1]) for i in range(1, len(cap) - 1, 2)}
    assert cap["high_water_keyframes"] <= 8 + 4 and cap["resident_now"] <= 8 and cap["evictions"] >= 40 and cap["mismatches"] == 0, cap
    wl = workloads.make_loop_closure_set("VLP-16", 3, n_candidates=5, n_distinct=3, downsample=0.4)
    wl.target.tofile(tmp_path / "t.bin")
    np.stack([L_colmajor(g) for g in wl.guesses]).astype(np.float32).tofile(tmp_path / "g.bin")
    files = []
    for i, c in enumerate(wl.candidates):
        c.tofile(tmp_path / f"c{i}.bin")
        files.append(str(tmp_path / f"c{i}.bin"))
    out = subprocess.run([exe, "0", str(n_engines), str(tmp_path / "t.bin"), str(tmp_path / "g.bin"), *files], check=True, capture_output=True,
                         text=True).stdout.splitlines()
    reg = select_registration_method({"registration_method": "FAST_GICP"})
    reg.setInputTarget(wl.target)
    clouds = [reg.upload(c) for c in wl.candidates]
    line = 0
    for order in (list(range(5)), [4, 3, 2, 1]):
        rec, best = reg.loop_match_batch([clouds[c] for c in order], [wl.guesses[c] for c in order], 4.0)
        head = out[line].split()
        assert head[0] == "best" and int(head[1]) == best and int(head[3]) == 5      # all five keyframes stay resident
        for k, c in enumerate(order):
            f = out[line + 1 + k].split()
            assert int(f[0]) == c and int(f[1]) == rec["converged"][k] and int(f[2]) == rec["iterations"][k]
            assert float(f[3]) == rec["fitness_score"][k]
            assert np.array_equal(np.array([float(v) for v in f[4:]], np.float32), rec["final_transformation"][k])
        line += 1 + len(order)
    reg.close()


def L_colmajor(T):
    from hdl_graph_slam_amd import _lib as L
    return L.colmajor16(T)


@pytest.mark.parametrize("method", [O.HGS_FAST_GICP, O.HGS_FAST_VGICP, O.HGS_NDT_OMP])
def test_api_call_sequences(method):
    """Orders of calls the nodelets can produce: source before target, one resident cloud as target AND source, a new target
    with the old source, back to a cached target, invalidation between two aligns, a batch whose candidate is the target."""
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.4)
    tgt2, src2, _ = synth.make_pair("VLP-16", 2, downsample=0.4)
    p = O.default_params(method)
    p.resolution = 1.0
    if method == O.HGS_NDT_OMP:
        p.max_iterations = 6       # keeps the six emulated NDT runs short; the sequence logic does not depend on it
    e, o = _engine(p), O.OracleRegistration(p)
    tol = dict(tol_m=1e-5, tol_rad=2e-5)
    e.setInputSource(src), o.setInputSource(src), e.setInputTarget(tgt), o.setInputTarget(tgt)
    PC.check_align(e, o, np.eye(4), **tol)
    c = e.upload(tgt)
    e.setInputTarget(c), e.setInputSource(c), o.setInputTarget(tgt), o.setInputSource(tgt)
    PC.check_align(e, o, synth.pose_matrix([0.05, 0.02, 0], [0, 0, 0.01]), **tol)
    PC.check_fitness(e, o, np.eye(4), max_ranges=(np.finfo(np.float64).max, 1.0))
    e.setInputTarget(tgt2), o.setInputTarget(tgt2), e.setInputSource(src2), o.setInputSource(src2)
    PC.check_align(e, o, np.eye(4), **tol)
    e.setInputTarget(c), o.setInputTarget(tgt), e.setInputSource(src), o.setInputSource(src)
    r1, _ = PC.check_align(e, o, np.eye(4), **tol)
    r2, _ = PC.check_align(e, o, np.eye(4), **tol)
    c.invalidate()
    r3, _ = PC.check_align(e, o, np.eye(4), **tol)
    assert bytes(r1.final_transformation) == bytes(r2.final_transformation) == bytes(r3.final_transformation)
    out = e.transformed_source(r1.matrix())
    ref = synth.xyz_of(src) @ r1.matrix()[:3, :3].T.astype(np.float32) + r1.matrix()[:3, 3].astype(np.float32)
    assert np.abs(synth.xyz_of(out) - ref).max() < 1e-4
    rec, best = e.loop_match_batch([c], [np.eye(4, dtype=np.float32)], 1.0)
    assert rec["converged"][0] == 1 and rec["fitness_score"][0] < 1e-4 and best == 0
    e.close()


def test_batch_records_do_not_depend_on_the_lane_count(monkeypatch):
    """HGS_BATCH_LANES (read in hgs_create) only changes how the problems are spread over streams."""
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    wl = workloads.make_loop_closure_set("VLP-16", 3, n_candidates=6, n_distinct=3, downsample=0.5)
    records = []
    for lanes in ("1", "4"):
        monkeypatch.setenv("HGS_BATCH_LANES", lanes)
        reg = select_registration_method({"registration_method": "FAST_GICP"})
        reg.setInputTarget(wl.target)
        rec, best = reg.loop_match_batch([reg.upload(c) for c in wl.candidates], wl.guesses, 4.0)
        records.append((rec.tobytes(), best))
        reg.close()
    assert records[0] == records[1]


def test_clouds_outlive_their_engine_safely():
    """hgs_destroy orphans the clouds the caller still holds: hgs_cloud_destroy frees the orphan, every other call rejects it."""
    import ctypes as C
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import HgsError
    tgt, src, T = _pair("vlp16")
    p = O.default_params(O.HGS_FAST_GICP)
    a, b = _engine(p), _engine(p)
    c = a.upload(src)
    assert c.size == len(src)
    a.close()
    assert L.lib().hgs_cloud_size(c._h) == 0
    out = np.zeros(4, np.float32)
    assert L.lib().hgs_cloud_download(c._h, out.ctypes.data_as(C.c_void_p), 16) == L.HGS_ERR_INVALID_ARGUMENT
    with pytest.raises(HgsError):
        b.setInputSource(c)
    c.close()
    b.close()


def test_long_ndt_run_ends_on_the_golden_float_matrix():
    """66 iterations of the emulated k_ndt_pass end bit for bit where the committed oracle trace (exact-sum mode) ends."""
    from test_golden import _check_long_ndt_run
    _check_long_ndt_run(_engine)


def test_sharded_batch_entry_point_world_one():
    """hgs_comm_init / hgs_loop_match_batch_sharded through the emulated library (its exchange step is a one-rank copy): record
    packing on the device, padding, scatter into candidate order, selection."""
    from test_distributed import _sharded_equals_unsharded
    _sharded_equals_unsharded(lambda: _engine(O.default_params(O.HGS_FAST_GICP)))


def test_golden_v3_next_rows():
    """deskewing, pcl::ApproximateVoxelGrid and the octree-ordered map cloud against the committed fixture."""
    from test_golden import _check_golden_v3
    e = _engine(O.default_params(O.HGS_FAST_GICP))
    _check_golden_v3(e)
    e.close()


def test_map_cloud_octree_growth():
    from test_map_cloud import _check_map_growth
    _check_map_growth(lambda: _engine(O.default_params(O.HGS_FAST_GICP)))


def test_deskewed_prefilter():
    """hgs_prefilter_deskewed (k_pf_load with the deskewing step) against the oracle, bit for bit."""
    from test_prefilter import _check_deskewed_prefilter
    _check_deskewed_prefilter(lambda: _engine(O.default_params(O.HGS_FAST_GICP)))


def test_approx_voxelgrid_in_eviction_order():
    """pcl::ApproximateVoxelGrid through the emulated k_pf_approx_* kernels: the oracle's sequential output, in order."""
    from test_prefilter import _check_approx_voxelgrid
    _check_approx_voxelgrid(lambda: _engine(O.default_params(O.HGS_FAST_GICP)))


def test_ndt_edge_cases():
    PC.check_ndt_edge_cases(_engine)


# ---- hgs_loop_match_batch_sharded with TWO ranks: two engines driven from two host threads of this process, the emulated
# communicator (tests/emul/simt_runtime.cpp) standing in for RCCL.  What runs is the C entry point itself: header gather, shard
# sizing, device-built records, padding, merge, and the rule that a rank whose own share is unusable still takes part.
def _two_ranks(world, body):
    import threading
    from hdl_graph_slam_amd.registration import RegistrationHIP
    uid = RegistrationHIP.comm_unique_id()
    out, errs = [None] * world, []

    def run(rank):
        try:
            out[rank] = body(rank, uid)
        except Exception as exc:  # noqa: BLE001
            errs.append((rank, repr(exc)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in th), "a rank is still blocked in the collective"
    assert not errs, errs
    return out


def test_sharded_batch_two_ranks_in_process(simt_library):
    from hdl_graph_slam_amd import _lib as L, workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    n_total, world = 5, 2
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=3, n_candidates=n_total, n_distinct=3, downsample=0.5)
    shards = [[0, 3, 4], [1, 2]]   # uneven on purpose: max(shard) slots travel, the smaller shard pads

    def body(rank, uid):
        reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
        reg.setInputTarget(wl.target)
        reg.comm_init(rank, world, uid)
        mine = shards[rank]
        rec, best = reg.loop_match_batch_sharded([reg.upload(wl.candidates[i]) for i in mine], mine, [wl.guesses[i] for i in mine], n_total, 4.0)
        reg.close()
        return rec.tobytes(), best
    res = _two_ranks(world, body)
    one = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
    one.setInputTarget(wl.target)
    rec, best = one.loop_match_batch([one.upload(c) for c in wl.candidates], wl.guesses, 4.0)
    rec["candidate_id"] = np.arange(n_total)
    one.close()
    for blob, b in res:
        assert blob == rec.tobytes() and b == best   # every rank: all records, candidate order, bits of the unsharded batch


def test_sharded_batch_rank_without_a_usable_share_does_not_block_the_others(simt_library):
    from hdl_graph_slam_amd import _lib as L, workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    n_total, world = 4, 2
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=4, n_candidates=n_total, n_distinct=2, downsample=0.5)

    def body(rank, uid):
        reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
        if rank == 0:
            reg.setInputTarget(wl.target)   # rank 1 never got the query keyframe: HGS_ERR_NO_TARGET, after the exchange
        reg.comm_init(rank, world, uid)
        mine = [i for i in range(n_total) if i % world == rank]
        rec, best, rc = reg.loop_match_batch_sharded([reg.upload(wl.candidates[i]) for i in mine], mine, [wl.guesses[i] for i in mine], n_total, 4.0,
                                                     return_status=True)
        reg.close()
        return rec.copy(), best, rc
    (rec0, best0, rc0), (rec1, best1, rc1) = _two_ranks(world, body)
    assert rc0 == L.HGS_OK and rc1 == L.HGS_ERR_NO_TARGET
    assert rec0.tobytes() == rec1.tobytes() and best0 == best1          # both hold the same merged list
    assert rec0["converged"][0::2].all() and not rec0["converged"][1::2].any()
    assert (rec0["fitness_score"][1::2] == L.DBL_MAX).all() and list(rec0["candidate_id"]) == list(range(n_total))
    assert best0 in (0, 2)


def test_sharded_batch_flags_a_candidate_two_ranks_report(simt_library):
    from hdl_graph_slam_amd import _lib as L, workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    n_total, world = 3, 2
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=4, n_candidates=n_total, n_distinct=2, downsample=0.5)
    shards = [[0, 1], [1, 2]]   # candidate 1 on both ranks: a broken partition

    def body(rank, uid):
        reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
        reg.setInputTarget(wl.target)
        reg.comm_init(rank, world, uid)
        mine = shards[rank]
        rec, best, rc = reg.loop_match_batch_sharded([reg.upload(wl.candidates[i]) for i in mine], mine, [wl.guesses[i] for i in mine], n_total, 4.0,
                                                     return_status=True)
        reg.close()
        return rc, bool(rec["converged"].all())
    res = _two_ranks(world, body)
    assert [r[0] for r in res] == [L.HGS_ERR_INVALID_ARGUMENT] * 2 and all(r[1] for r in res)


def _rank_fails_after_the_header(kind):
    from hdl_graph_slam_amd import _lib as L, workloads
    from hdl_graph_slam_amd.registrations import select_registration_method
    n_total, world = 4, 2
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=4, n_candidates=n_total, n_distinct=2, downsample=0.5)

    def body(rank, uid):
        reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
        reg.setInputTarget(wl.target)
        reg.comm_init(rank, world, uid)
        mine = [i for i in range(n_total) if i % world == rank]
        rec, best, rc = reg.loop_match_batch_sharded([reg.upload(wl.candidates[i]) for i in mine], mine, [wl.guesses[i] for i in mine], n_total, 4.0,
                                                     return_status=True)
        reg.close()
        return rec.copy(), best, rc
    os.environ["HGS_FAULT_AFTER_HEADER"] = f"{kind}:1"   # rank 1 fails between the header gather and the record gather
    try:
        return L, _two_ranks(world, body)
    finally:
        del os.environ["HGS_FAULT_AFTER_HEADER"]


def test_sharded_batch_rank_that_throws_in_its_batch_after_the_header_still_sends_padding(simt_library):
    """A C++ exception on one rank where its batch runs (the peers already know its shard size): the rank takes part in the record
    all-gather with padding and reports after the exchange; the other rank returns HGS_OK with that rank's candidates not converged."""
    L, ((rec0, best0, rc0), (rec1, best1, rc1)) = _rank_fails_after_the_header("batch")
    assert rc0 == L.HGS_OK and rc1 == L.HGS_ERR_OUT_OF_MEMORY
    assert rec0.tobytes() == rec1.tobytes() and best0 == best1
    assert rec0["converged"][0::2].all() and not rec0["converged"][1::2].any() and best0 in (0, 2)


def test_sharded_batch_rank_that_leaves_between_the_collectives_aborts_the_communicator(simt_library):
    """An exception outside every handler between the two collectives: the scope guard aborts the communicator, nobody blocks (the
    thread join of _two_ranks), the peer's collective fails with HGS_ERR_COMM instead of waiting for records that never come."""
    L, ((_, _, rc0), (_, _, rc1)) = _rank_fails_after_the_header("guard")
    assert rc0 == L.HGS_ERR_COMM and rc1 == L.HGS_ERR_INTERNAL


def test_ndt_pass_with_one_block_and_long_runs_of_one_problem(simt_library):
    """k_ndt_pass's LDS lane slots hold at most kNdtFlushTiles tiles of one problem between flushes (their integer sums must stay below 2^56): with ONE
    resident block a 12 k-point cloud is 48 consecutive tiles of one problem on that block, so the forced mid-run flush runs — and the result is still
    the oracle's exact-sum result bit for bit (the sums do not depend on how the tiles were grouped)."""
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.18)
    assert len(src) > 33 * 256, len(src)
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution, p.neighbor_search = 1.0, O.HGS_DIRECT7
    o = O.OracleRegistration(p).set_ndt_sum_mode(1)
    o.setInputTarget(tgt)
    o.setInputSource(src)
    guess = T @ synth.pose_matrix([0.05, -0.03, 0.0], [0.0, 0.0, 0.004])
    ro = o.align(guess)
    for resident in ("1", "3"):
        os.environ["HGS_NDT_RESIDENT"] = resident
        try:
            e = _engine(p)
        finally:
            del os.environ["HGS_NDT_RESIDENT"]
        e.setInputTarget(tgt)
        e.setInputSource(src)
        r = e.align(guess)
        assert r.iterations == ro.iterations and bytes(r.final_transformation) == bytes(ro.final_transformation), (resident, r.iterations, ro.iterations)
        e.close()


def test_sharded_batch_rank_whose_peer_never_enters_gives_up_after_the_deadline(simt_library):
    """One rank calls the collective, the other never does (a SLAM process that died before the detection): the caller must come back with
    HGS_ERR_COMM after HGS_COMM_TIMEOUT_MS instead of waiting for ever, and its engine stays usable for unsharded work."""
    from hdl_graph_slam_amd import _lib as L, workloads
    from hdl_graph_slam_amd.registration import RegistrationHIP
    from hdl_graph_slam_amd.registrations import select_registration_method
    import time
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=4, n_candidates=2, n_distinct=2, downsample=0.5)
    uid = RegistrationHIP.comm_unique_id()
    reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
    reg.setInputTarget(wl.target)
    reg.comm_init(0, 2, uid)   # world size 2, the second rank never joins the collective
    os.environ["HGS_COMM_TIMEOUT_MS"] = "300"
    try:
        t0 = time.time()
        rec, best, rc = reg.loop_match_batch_sharded([reg.upload(wl.candidates[0])], [0], [wl.guesses[0]], 2, 4.0, return_status=True)
        waited = time.time() - t0
    finally:
        del os.environ["HGS_COMM_TIMEOUT_MS"]
    assert rc == L.HGS_ERR_COMM and 0.25 < waited < 30.0, (rc, waited)
    rec1, best1 = reg.loop_match_batch([reg.upload(c) for c in wl.candidates], wl.guesses, 4.0)   # the engine itself is fine
    assert rec1["converged"].all()
    reg.close()


def test_nn1_with_equidistant_target_points(simt_library):
    """The quad walk's min-only leaves + resolve + keyed-walk fallback on duplicated / regular-grid targets (PC docstring)."""
    PC.check_nn1_with_equidistant_targets(_engine)


def test_nn1_on_small_trees(simt_library):
    """Quad walk on trees of 3 ... 129 leaves, odd and even heights (PC docstring)."""
    PC.check_nn1_on_small_trees(_engine)
