"""Builds and loads tests/emul/libhgs_simt.so: the PRODUCT sources (hdl_graph_slam_amd/csrc/hgs_kernels.hip and
hgs_engine.hip, unchanged) compiled for the host against the SIMT emulation shim (tests/emul/simt/hip/hip_runtime.h), so that
the kernels and the engine behind the C-ABI run on the CPU — GPU threads as fibers, waves meeting in the cross-lane
operations.  Test infrastructure: nothing in hdl_graph_slam_amd/ knows about it."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "hdl_graph_slam_amd", "csrc")
LIB = os.path.join(HERE, "libhgs_simt.so")


def _clang() -> str | None:
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    return None


def build() -> str | None:
    """Returns the library path, or None when no clang++ is available (ext_vector_type / elementwise builtins need clang)."""
    cxx = _clang()
    if cxx is None:
        return None
    srcs = [os.path.join(CSRC, "hgs_kernels.hip"), os.path.join(CSRC, "hgs_engine.hip"), os.path.join(HERE, "simt_runtime.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(HERE, "simt", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "hgs_registration.h"), os.path.abspath(__file__)]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    objs = []
    for src in srcs:
        obj = os.path.join(HERE, "_simt_" + os.path.basename(src) + ".o")
        cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-fno-strict-aliasing", "-ffp-contract=off", "-Wno-unknown-attributes",
               "-Wno-unused-value", "-Wno-pass-failed", "-DHGS_TESTING", "-I", os.path.join(HERE, "simt"), "-c", src, "-o", obj]
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run([cxx, "-shared", "-o", LIB + ".tmp", *objs], check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB
