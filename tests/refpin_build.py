"""TEST INFRASTRUCTURE: builds tests/_refpin/refpin_main = tests/cpp/refpin_main.cpp + two UNMODIFIED reference translation units,
compiled from where they lie under /root/reference (no reference source enters this repository):

    src/hdl_graph_slam/information_matrix_calculator.cpp      calc_fitness_score (:49-80), calc_information_matrix (:25-47)
    src/hdl_graph_slam/keyframe.cpp                           KeyFrame::save (:21-58), KeyFrame::load (:60-145)

against the stand-in ROS / PCL / Eigen / g2o / boost headers of tests/mock_* (none of the real ones exist in this image).  This is NOT an
`oracle/_ref` build of the reference's engines (ndt_omp / fast_gicp / PCL are absent and unbuildable): it runs the only two
pieces of the hot path's arithmetic / data format that live in the reference tree itself, so that `oracle/`, `keyframe_io.py` and the HIP path
can be checked against reference code executed here instead of against our restatement alone.  The binary is git-ignored (tests/_refpin/)
and travels to the GPU box like the other built artefacts; the vectors it produced are committed under tests/golden/ (make_refpin_golden.py).
-ffp-contract=off: the stand-in for pcl::transformPointCloud states PCL's unfused mul / add order."""
from __future__ import annotations

import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
UNITS = ["src/hdl_graph_slam/information_matrix_calculator.cpp", "src/hdl_graph_slam/keyframe.cpp"]
HEADERS = ["include/hdl_graph_slam/information_matrix_calculator.hpp", "include/hdl_graph_slam/keyframe.hpp"]
OUT = os.path.join(ROOT, "tests", "_refpin")
DRIVER = os.path.join(ROOT, "tests", "cpp", "refpin_main.cpp")


def have_reference() -> bool:
    return all(os.path.exists(os.path.join(REFERENCE, f)) for f in UNITS + HEADERS)


def exe(variant: str = "") -> str:
    return os.path.join(OUT, "refpin_main" + (f"_{variant}" if variant else ""))


def _deps():
    d = [DRIVER, os.path.abspath(__file__)] + [os.path.join(REFERENCE, f) for f in UNITS + HEADERS]
    for mock in ("mock_ros", "mock_pcl", "mock_eigen"):
        for base, _, files in os.walk(os.path.join(ROOT, "tests", mock)):
            d += [os.path.join(base, f) for f in files]
    return d


def build(variant: str = "") -> str | None:
    """variant "": pcl::transformPointCloud in PCL >= 1.10's order (noetic); "pcl18": PCL 1.8's (melodic).  Returns the executable, or None
    when the reference tree is absent and no prebuilt binary exists."""
    out = exe(variant)
    if not have_reference():
        return out if os.path.exists(out) else None
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in _deps()):
        return out
    os.makedirs(OUT, exist_ok=True)
    inc = []
    for d in (os.path.join(ROOT, "tests", "mock_ros"), os.path.join(ROOT, "tests", "mock_pcl"), os.path.join(ROOT, "tests", "mock_eigen"),
              os.path.join(REFERENCE, "include")):
        inc += ["-I", d]
    flags = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", *inc] + (["-DHGS_MOCK_PCL_1_8"] if variant == "pcl18" else [])
    objs = []
    for i, src in enumerate([os.path.join(REFERENCE, u) for u in UNITS] + [DRIVER]):
        obj = os.path.join(OUT, f"unit{i}{'_' + variant if variant else ''}.o")
        subprocess.run([*flags, "-c", src, "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["g++", *objs, "-o", out + ".tmp"], check=True)
    os.replace(out + ".tmp", out)
    return out


def run(args, variant: str = "") -> str:
    e = build(variant)
    if e is None:
        raise RuntimeError("refpin_main is not available (no /root/reference and no prebuilt binary)")
    return subprocess.run([e, *[str(a) for a in args]], check=True, capture_output=True, text=True).stdout


if __name__ == "__main__":
    print(build(), build("pcl18"))
