"""TEST INFRASTRUCTURE: applies integration/hdl_graph_slam_hip.patch to a scratch copy of the reference files it touches and compiles
the patched src/hdl_graph_slam/registrations.cpp + a driver that includes the patched include/hdl_graph_slam/loop_detector.hpp
(tests/cpp/integration_main.cpp) against the stand-in headers of tests/mock_ros, tests/mock_pcl, tests/mock_eigen.

The reference tree exists only in the build container (/root/reference): the binaries land in integration/_build/ (git-ignored, like every
built artefact; NOT gpurun-ignored, so they travel to the GPU box the way the .so files do) and the `-m gpu` test runs the prebuilt one.
The patched scratch copy lives in a temporary directory and is deleted: no reference source enters this repository."""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "hdl_graph_slam_hip.patch")
OUT = os.path.join(ROOT, "integration", "_build")
PATCHED = ["CMakeLists.txt", "src/hdl_graph_slam/registrations.cpp", "include/hdl_graph_slam/loop_detector.hpp", "apps/scan_matching_odometry_nodelet.cpp",
           "src/hdl_graph_slam/information_matrix_calculator.cpp", "src/hdl_graph_slam/map_cloud_generator.cpp", "apps/prefiltering_nodelet.cpp"]
UNTOUCHED = ["include/hdl_graph_slam/registrations.hpp", "include/hdl_graph_slam/keyframe.hpp", "include/hdl_graph_slam/graph_slam.hpp", "include/hdl_graph_slam/ros_utils.hpp",
             "include/hdl_graph_slam/information_matrix_calculator.hpp", "include/hdl_graph_slam/map_cloud_generator.hpp"]
# patched translation units that are compiled and linked into integration_main.  The two patched nodelets (apps/*.cpp: the classes live in the .cpp files) are
# compiled too — prefilter_nodelet_main, odometry_nodelet_main include them — against the stand-in ROS graph / messages / tf / PCL filters of tests/mock_*, and
# their onInit / cloud_callback / matching / publish_scan_matching_status RUN
COMPILED = ["src/hdl_graph_slam/registrations.cpp", "src/hdl_graph_slam/information_matrix_calculator.cpp", "src/hdl_graph_slam/map_cloud_generator.cpp"]


def have_reference() -> bool:
    return all(os.path.exists(os.path.join(REFERENCE, f)) for f in PATCHED + UNTOUCHED)


def exe(kind: str, name: str = "integration_main") -> str:
    """name: "integration_main" (factory, loop detector, f1, f3, the f2 adapter call), "prefilter_nodelet_main" / "odometry_nodelet_main" (the patched
    apps/prefiltering_nodelet.cpp / apps/scan_matching_odometry_nodelet.cpp themselves, run against the stand-in ROS graph)"""
    return os.path.join(OUT, name + ("_simt" if kind == "simt" else ""))


def _deps(lib):
    d = [PATCH, lib, os.path.join(ROOT, "tests", "cpp", "integration_main.cpp"), os.path.join(ROOT, "tests", "cpp", "prefilter_nodelet_main.cpp"),
         os.path.join(ROOT, "tests", "cpp", "odometry_nodelet_main.cpp"), os.path.join(ROOT, "oracle", "prefilter.hpp"), os.path.join(ROOT, "adapters", "registration_hip.hpp"),
         os.path.join(ROOT, "adapters", "loop_match_hip.hpp"), os.path.join(ROOT, "adapters", "resident_clouds_hip.hpp"), os.path.join(ROOT, "include", "hgs_registration.h"),
         os.path.join(ROOT, "oracle", "mapcloud.hpp"), os.path.abspath(__file__)]
    for mock in ("mock_ros", "mock_pcl", "mock_eigen"):
        for base, _, files in os.walk(os.path.join(ROOT, "tests", mock)):
            d += [os.path.join(base, f) for f in files]
    return d


def apply_patch(dst: str) -> None:
    """Copies the files the patch touches (+ the three untouched headers the patched ones include) into dst and applies the patch there."""
    for f in PATCHED + UNTOUCHED:
        os.makedirs(os.path.dirname(os.path.join(dst, f)), exist_ok=True)
        shutil.copy(os.path.join(REFERENCE, f), os.path.join(dst, f))
    subprocess.run(["git", "apply", "--whitespace=nowarn", PATCH], cwd=dst, check=True)


def build(kind: str = "hip") -> str | None:
    """kind "hip": linked with hdl_graph_slam_amd/lib/libhgs_hip.so (runs on the GPU box); "simt": with tests/emul/libhgs_simt.so (the kernels
    emulated on the host).  Returns the executable, or None when it cannot be (re)built here and no prebuilt one exists."""
    if kind == "simt":
        from emul import simt
        lib = simt.build()
        if lib is None:
            return None
    else:
        from hdl_graph_slam_amd import build as hip_build
        lib = hip_build.build_lib()
    out, out_nodelet, out_odometry = exe(kind), exe(kind, "prefilter_nodelet_main"), exe(kind, "odometry_nodelet_main")
    if not have_reference():
        return out if all(os.path.exists(o) for o in (out, out_nodelet, out_odometry)) else None
    if all(os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in _deps(lib)) for o in (out, out_nodelet, out_odometry)):
        return out
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        apply_patch(tmp)
        inc = []
        for d in (os.path.join(ROOT, "tests", "mock_ros"), os.path.join(ROOT, "tests", "mock_pcl"), os.path.join(ROOT, "tests", "mock_eigen"),
                  os.path.join(tmp, "include"), os.path.join(ROOT, "include"), os.path.join(ROOT, "adapters")):
            inc += ["-I", d]
        flags = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unknown-pragmas", "-DUSE_HGS_HIP", *inc]
        objs = []
        for i, unit in enumerate(COMPILED):
            objs.append(os.path.join(tmp, f"unit{i}.o"))
            subprocess.run([*flags, "-c", os.path.join(tmp, unit), "-o", objs[-1]], check=True)
            # every patched unit must also still build WITHOUT the backend (the patch is all #ifdef USE_HGS_HIP)
            subprocess.run([f for f in flags if f != "-DUSE_HGS_HIP"] + ["-fsyntax-only", os.path.join(tmp, unit)], check=True)
        # ... and the UNPATCHED units, under other symbol names, so that the tests can run the reference's CPU code next to the device path
        for i, (unit, rename) in enumerate((("src/hdl_graph_slam/information_matrix_calculator.cpp", "InformationMatrixCalculator=InformationMatrixCalculatorCPU"),
                                            ("src/hdl_graph_slam/map_cloud_generator.cpp", "MapCloudGenerator=MapCloudGeneratorCPU"))):
            objs.append(os.path.join(tmp, f"cpu{i}.o"))
            subprocess.run([f for f in flags if f != "-DUSE_HGS_HIP"] + ["-ffp-contract=off", f"-D{rename}", "-c", os.path.join(REFERENCE, unit), "-o", objs[-1]], check=True)
        objs.append(os.path.join(tmp, "main.o"))
        subprocess.run([*flags, "-c", os.path.join(ROOT, "tests", "cpp", "integration_main.cpp"), "-o", objs[-1]], check=True)
        libdir = os.path.dirname(lib)
        rpath = "$ORIGIN/" + os.path.relpath(libdir, OUT)
        link = ["-l:libhgs_simt.so"] if kind == "simt" else ["-lhgs_hip"]
        subprocess.run(["g++", *objs, "-o", out + ".tmp", "-L", libdir, *link, "-pthread", f"-Wl,-rpath,{rpath}"], check=True)
        os.replace(out + ".tmp", out)
        # the patched prefiltering nodelet: the class lives in the .cpp, the driver includes it (the stand-in ROS graph of tests/mock_ros lets onInit / cloud_callback run);
        # it must also still compile without the backend
        nodelet_flags = [*flags, "-I", tmp]
        subprocess.run([*nodelet_flags, "-c", os.path.join(ROOT, "tests", "cpp", "prefilter_nodelet_main.cpp"), "-o", os.path.join(tmp, "nodelet.o")], check=True)
        subprocess.run([f for f in flags if f != "-DUSE_HGS_HIP"] + ["-fsyntax-only", os.path.join(tmp, "apps", "prefiltering_nodelet.cpp")], check=True)
        subprocess.run(["g++", os.path.join(tmp, "nodelet.o"), "-o", out_nodelet + ".tmp", "-L", libdir, *link, "-pthread", f"-Wl,-rpath,{rpath}"], check=True)
        os.replace(out_nodelet + ".tmp", out_nodelet)
        # the patched odometry nodelet + the patched factory: the reference's own ScanMatchingOdometryNodelet drives the engine
        subprocess.run([*nodelet_flags, "-Wno-unused-function", "-c", os.path.join(ROOT, "tests", "cpp", "odometry_nodelet_main.cpp"), "-o", os.path.join(tmp, "odometry.o")], check=True)
        subprocess.run([f for f in flags if f != "-DUSE_HGS_HIP"] + ["-Wno-unused-function", "-fsyntax-only", os.path.join(tmp, "apps", "scan_matching_odometry_nodelet.cpp")], check=True)
        subprocess.run(["g++", os.path.join(tmp, "odometry.o"), objs[0], "-o", out_odometry + ".tmp", "-L", libdir, *link, "-pthread", f"-Wl,-rpath,{rpath}"], check=True)
        os.replace(out_odometry + ".tmp", out_odometry)
    return out
