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
