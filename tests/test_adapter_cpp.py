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
    deps = [src, os.path.join(ROOT, "adapters", "registration_hip.hpp"), os.path.join(ROOT, "include", "hgs_registration.h"), lib]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "mock_pcl"), "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                        "-L", os.path.dirname(lib), "-lhgs_hip", f"-Wl,-rpath,{os.path.dirname(lib)}"], check=True)
    return EXE


def test_adapter_compiles_and_links():
    assert os.path.exists(_build())


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
    reg.close()
