"""The header-only pcl::Registration adapter (adapters/registration_hip.hpp) — what the reference's factory would
construct — compiled against the mock PCL of tests/mock_pcl and linked with libhgs_hip.so.
CPU: it compiles and links (the adapter only uses the C-ABI of include/hgs_registration.h and the pcl::Registration base
members the callers rely on).  GPU: driven like scan_matching_odometry_nodelet.cpp:166-221 it returns the pose the Python
mirror returns."""
import os
import subprocess

import numpy as np
import pytest

from hdl_graph_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_main")


def _build():
    from hdl_graph_slam_amd import build as hip_build
    lib = hip_build.build_lib()
    src = os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp")
    deps = [src, os.path.join(ROOT, "adapters", "registration_hip.hpp"), os.path.join(ROOT, "include", "hgs_registration.h"), lib,
            os.path.join(ROOT, "tests", "mock_pcl", "pcl", "registration", "registration.h"), os.path.join(ROOT, "tests", "mock_pcl", "pcl", "search", "kdtree.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "mock_pcl"), "-I", os.path.join(ROOT, "tests", "mock_eigen"), "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                        "-L", os.path.dirname(lib), "-lhgs_hip", f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    return EXE


def test_adapter_compiles_and_links():
    assert os.path.exists(_build())


def test_adapter_compiles_against_the_pcl_1_12_signatures():
    """pcl::search::KdTree::setInputCloud returns bool from PCL 1.12 on (void before): the adapter's LazyKdTree overrides it under
    PCL_VERSION_COMPARE — the stand-in switches signature and version with -DHGS_MOCK_PCL_1_12 so that the other branch is compiled too."""
    src = os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror=overloaded-virtual", "-fsyntax-only", "-DHGS_MOCK_PCL_1_12", "-I", os.path.join(ROOT, "tests", "mock_pcl"),
                    "-I", os.path.join(ROOT, "tests", "mock_eigen"), "-I", os.path.join(ROOT, "include"), src], check=True)


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 2])
def test_adapter_matches_python_mirror(tmp_path, method):
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    exe = _build()
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.3)
    tgt.tofile(tmp_path / "t.bin")
    src.tofile(tmp_path / "s.bin")
    out = subprocess.run([exe, str(method), str(tmp_path / "t.bin"), str(tmp_path / "s.bin")], check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "converged 1"
    Tc = np.array([float(v) for v in out[1].split()], np.float32).reshape(4, 4).T
    p = L.default_params(method)
    if method == L.HGS_NDT_OMP:
        p.resolution, p.neighbor_search = 1.0, L.HGS_DIRECT7
    reg = RegistrationHIP(p)
    reg.setInputTarget(tgt)
    reg.setInputSource(src)
    r = reg.align(np.eye(4))
    assert np.array_equal(Tc, r.matrix())                       # same library, same inputs: identical bits
    assert abs(float(out[2].split()[1]) - reg.getFitnessScore()) < 1e-9      # printed with 12 significant digits
    assert out[4] == "no_device converged 0 guess_kept 1"              # hgs_create failure: no exception, hasConverged() false, guess kept
    assert out[5] == "recovered converged 1 same_pose 1"                # the engine created later holds the clouds set while creation failed
    from parity_checks import check_adapter_lazy_tree_lines
    check_adapter_lazy_tree_lines(out)
    reg.close()


@pytest.mark.gpu
def test_cpp_loop_matcher_on_the_gpu(tmp_path):
    """adapters/loop_match_hip.hpp (the replacement of the loop body of LoopDetector::matching, loop_detector.hpp:135-154)
    compiled against the REAL libhgs_hip.so and run on the GPU: two detections over resident keyframes, against the Python
    mirror's batches, bit for bit."""
    from hdl_graph_slam_amd import build as hip_build, workloads, _lib as L
    from hdl_graph_slam_amd.registrations import select_registration_method
    lib = hip_build.build_lib()
    exe = os.path.join(ROOT, "tests", "cpp", "loop_match_main")
    src_cpp = os.path.join(ROOT, "tests", "cpp", "loop_match_main.cpp")
    deps = [src_cpp, os.path.join(ROOT, "adapters", "loop_match_hip.hpp"), os.path.join(ROOT, "include", "hgs_registration.h"), lib]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), src_cpp, "-o", exe, "-L", os.path.dirname(lib), "-lhgs_hip",
                        f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    wl = workloads.make_loop_closure_set("VLP-16", 3, n_candidates=5, n_distinct=3, downsample=0.4)
    wl.target.tofile(tmp_path / "t.bin")
    np.stack([L.colmajor16(g) for g in wl.guesses]).astype(np.float32).tofile(tmp_path / "g.bin")
    files = []
    for i, c in enumerate(wl.candidates):
        c.tofile(tmp_path / f"c{i}.bin")
        files.append(str(tmp_path / f"c{i}.bin"))
    out = subprocess.run([exe, "0", "1", str(tmp_path / "t.bin"), str(tmp_path / "g.bin"), *files], check=True, capture_output=True, text=True).stdout.splitlines()
    reg = select_registration_method({"registration_method": "FAST_GICP"})
    reg.setInputTarget(wl.target)
    clouds = [reg.upload(c) for c in wl.candidates]
    line = 0
    for order in (list(range(5)), [4, 3, 2, 1]):
        rec, best = reg.loop_match_batch([clouds[c] for c in order], [wl.guesses[c] for c in order], 4.0)
        head = out[line].split()
        assert head[0] == "best" and int(head[1]) == best and int(head[3]) == 5
        for k, c in enumerate(order):
            f = out[line + 1 + k].split()
            assert int(f[0]) == c and int(f[1]) == rec["converged"][k] and int(f[2]) == rec["iterations"][k]
            assert float(f[3]) == rec["fitness_score"][k]
            assert np.array_equal(np.array([float(v) for v in f[4:]], np.float32), rec["final_transformation"][k])
        line += 1 + len(order)
    reg.close()


def _capacity_line(exe, *args):
    out = subprocess.run([exe, "capacity", *[str(a) for a in args]], check=True, capture_output=True, text=True).stdout.split()
    return {out[i]: int(out[i + 1]) for i in range(1, len(out) - 1, 2)}


@pytest.mark.gpu
def test_resident_keyframes_stay_bounded_over_a_2000_keyframe_run():
    """hdl_graph_slam never removes a keyframe: LoopMatcherHIP's budget (reg_hip_resident_keyframes / reg_hip_resident_mb of the patched LoopDetector) keeps the
    device copies bounded, least recently matched first — 2000 keyframes, four per detection + one old one, at most 32 resident / 24 MiB; a keyframe that was
    evicted and uploaded again gives the same record bit for bit."""
    from hdl_graph_slam_amd import build as hip_build
    lib = hip_build.build_lib()
    exe = os.path.join(ROOT, "tests", "cpp", "loop_match_main")
    src_cpp = os.path.join(ROOT, "tests", "cpp", "loop_match_main.cpp")
    deps = [src_cpp, os.path.join(ROOT, "adapters", "loop_match_hip.hpp"), os.path.join(ROOT, "include", "hgs_registration.h"), lib]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), src_cpp, "-o", exe, "-L", os.path.dirname(lib), "-lhgs_hip",
                        f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    r = _capacity_line(exe, 0, 2000, 4, 32, 0, 2000)          # count budget
    assert r["detections"] == 500 and r["high_water_keyframes"] <= 32 + 5 and r["resident_now"] <= 32 and r["evictions"] >= 1900 and r["mismatches"] == 0, r
    r = _capacity_line(exe, 0, 2000, 4, 0, 24, 2000)           # byte budget: 24 MiB
    assert r["high_water_bytes"] <= 24 * 2 ** 20 + 5 * 2 ** 20 and r["evictions"] >= 1500 and r["mismatches"] == 0, r
    r = _capacity_line(exe, 0, 200, 4, 0, 0, 2000)             # no budget: everything stays
    assert r["resident_now"] == 200 and r["evictions"] == 0 and r["mismatches"] == 0, r
