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


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(_HERE, "..", "include", "hgs_registration.h")]


def _obj_stale(src: str) -> bool:
    """An object is stale when it is older than ITS source or any header (not than the other .hip files: an edit to one
    source must not make the untouched objects look stale for ever)."""
    obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + _headers())


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if any(_obj_stale(src) for src in SOURCES):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(LIB_DIR, src.replace(".hip", ".o"))) > t for src in SOURCES)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        if force or _obj_stale(src):
            cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(obj)
    # no -lrccl: hgs_comm.hip loads RCCL with dlopen at the first hgs_comm_* call, the registration path does not depend on it.
    # Linked next to the target and renamed into place: a process that has the old library mapped keeps its (unlinked) file, and
    # ranks / tests that start together never see a half-written one.
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
