"""Builds hdl_graph_slam_amd/lib/libhgs_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhgs_hip.so")
SOURCES = ["hgs_sort.hip", "hgs_kernels.hip", "hgs_engine.hip", "hgs_comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X backend cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    objs = [os.path.join(LIB_DIR, src.replace(".hip", ".o")) for src in SOURCES]
    if not all(os.path.exists(o) for o in objs):
        return True
    t = min([os.path.getmtime(LIB_PATH)] + [os.path.getmtime(o) for o in objs])   # the oldest object decides: a relink alone is not a rebuild
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "hgs_registration.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        src_path = os.path.join(CSRC, src)
        deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f == src] + [os.path.join(_HERE, "..", "include", "hgs_registration.h")]
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            cmd = [hipcc, *FLAGS, "-c", src_path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(obj)
    # no -lrccl: hgs_comm.hip loads RCCL with dlopen at the first hgs_comm_* call, the registration path does not depend on it
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
