"""TEST-ONLY: ctypes front-end of tests/emul/libhgs_emul.so (host execution of the HGS_HD device functions)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhgs_emul.so")
_ROOT = os.path.dirname(os.path.dirname(_HERE))


def build():
    srcs = [os.path.join(_HERE, "emul.cpp")] + [os.path.join(_ROOT, "hdl_graph_slam_amd", "csrc", f) for f in ("hgs_math.h", "hgs_bvh.h", "hgs_gicp.h", "hgs_ndt.h", "hgs_vgicp.h")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-mfma", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                        "-o", _LIB, srcs[0]], check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.emul_create.restype = vp
        L.emul_create.argtypes = [vp]
        L.emul_destroy.argtypes = [vp]
        L.emul_set_target.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.emul_set_source.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
        L.emul_align.argtypes = [vp, vp, vp]
        L.emul_fitness.argtypes = [vp, vp, C.c_double, vp, vp]
        L.emul_nn_target.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, vp]
        L.emul_target_covariances.argtypes = [vp, vp]
        L.emul_gicp_linearize.argtypes = [vp, vp, vp, vp, vp, vp]
        L.emul_ndt_cells.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.emul_ndt_grid.argtypes = [vp, vp, vp]
        L.emul_ndt_derivatives.argtypes = [vp, vp, vp, vp, vp]
        L.emul_sorted_order.argtypes = [vp, C.c_int, vp]
        L.emul_walk_stats.argtypes = [vp, vp, C.c_float, C.c_int, vp]
        L.emul_walk_stats_knn.argtypes = [vp, C.c_int, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class EmulRegistration:
    def __init__(self, params):
        import oracle as O
        self.params = params
        self._h = lib().emul_create(C.byref(params))
        self._O = O
        self.result = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().emul_destroy(self._h)
            self._h = None

    def setInputTarget(self, cloud):
        c, n, s = self._O._cloud_args(cloud)
        self.n_target = n
        lib().emul_set_target(self._h, _p(c), n, s)

    def setInputSource(self, cloud):
        c, n, s = self._O._cloud_args(cloud)
        self.n_source = n
        lib().emul_set_source(self._h, _p(c), n, s)

    def align(self, guess=None):
        g, _ = self._O.colmajor16(np.eye(4) if guess is None else guess)
        r = self._O.HgsResult()
        lib().emul_align(self._h, _p(g), C.byref(r))
        self.result = r
        return r

    def getFinalTransformation(self):
        return self.result.matrix()

    def getFitnessScore(self, max_range=np.finfo(np.float64).max, T=None):
        g, _ = self._O.colmajor16(self.getFinalTransformation() if T is None else T)
        s, n = C.c_double(), C.c_uint32()
        lib().emul_fitness(self._h, _p(g), max_range, C.byref(s), C.byref(n))
        self.last_num_inliers = n.value
        return s.value

    def nn_target(self, q):
        q = np.ascontiguousarray(q, np.float32)
        idx, d2 = np.empty(len(q), np.int32), np.empty(len(q), np.float32)
        lib().emul_nn_target(self._h, _p(q), len(q), q.strides[0], _p(idx), _p(d2))
        return idx, d2

    def target_covariances(self, n=None):
        out = np.zeros((self.n_target, 6), np.float32)
        lib().emul_target_covariances(self._h, _p(out))
        return out

    def gicp_linearize(self, T):
        T12 = np.ascontiguousarray(np.asarray(T, np.float64)[:3, :4])
        H, b, e = np.zeros((6, 6)), np.zeros(6), np.zeros(1)
        corr = np.empty(self.n_source, np.int32)
        lib().emul_gicp_linearize(self._h, _p(T12), _p(H), _p(b), _p(e), _p(corr))
        return H, b, float(e[0]), corr

    def ndt_cells(self):
        cap = 1 << 20
        key, mean, icov, npts = np.zeros(cap, np.int32), np.zeros((cap, 3)), np.zeros((cap, 6), np.float32), np.zeros(cap, np.int32)
        n = lib().emul_ndt_cells(self._h, cap, _p(key), _p(mean), _p(icov), _p(npts))
        min_b, mul = np.zeros(3, np.int32), np.zeros(3, np.int32)
        lib().emul_ndt_grid(self._h, _p(min_b), _p(mul))
        key = key[:n]
        ijk = np.stack([key % mul[1], (key // mul[1]) % (mul[2] // mul[1]), key // mul[2]], 1) + min_b
        return ijk, mean[:n].copy(), icov[:n].copy(), npts[:n].copy()

    def ndt_derivatives(self, p6):
        p = np.ascontiguousarray(p6, np.float64)
        s, g, H = np.zeros(1), np.zeros(6), np.zeros((6, 6))
        lib().emul_ndt_derivatives(self._h, _p(p), _p(s), _p(g), _p(H))
        return float(s[0]), g, H

    def sorted_order(self, target=True):
        out = np.zeros(self.n_target if target else self.n_source, np.int32)
        n = lib().emul_sorted_order(self._h, int(target), _p(out))
        return out[:n]

    def walk_stats(self, T, bound2=np.finfo(np.float32).max, use_seed=False):
        """Host simulation of the wave-cooperative 4-ary walk (hgs_wave_bvh.h) over the source cloud at pose T:
        (group evaluations, leaf visits, waves, mismatches vs the per-lane search)."""
        out = np.zeros(5, np.int64)
        T16 = np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(-1))
        lib().emul_walk_stats(self._h, _p(T16), C.c_float(bound2), int(use_seed), _p(out))
        return tuple(int(v) for v in out[:4])

    def walk_stats_knn(self, k=20):
        out = np.zeros(5, np.int64)
        lib().emul_walk_stats_knn(self._h, int(k), _p(out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[4])
