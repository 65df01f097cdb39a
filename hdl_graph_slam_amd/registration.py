"""pcl::Registration-shaped front-end of the MI355X backend.

Mirrors the surface hdl_graph_slam's callers use on the object returned by select_registration_method
(reference: apps/scan_matching_odometry_nodelet.cpp:172-246,298-335 and
include/hdl_graph_slam/loop_detector.hpp:122-153): setInputTarget / setInputSource / align / hasConverged /
getFinalTransformation / getFitnessScore / getSearchMethodTarget()->nearestKSearch.  Every call goes through the
C-ABI of include/hgs_registration.h into the HIP library; nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L


class HgsError(RuntimeError):
    pass


class DeviceCloud:
    """A cloud resident in HBM (hgs_cloud): a KeyFrame::cloud uploaded once and reused as loop-closure candidate."""

    def __init__(self, reg: "RegistrationHIP", cloud: np.ndarray):
        arr, n, stride = L.cloud_args(cloud)
        self._reg = reg
        self._h = C.c_void_p()
        reg._check(L.lib().hgs_cloud_create(reg._h, arr.ctypes.data_as(C.c_void_p), n, stride, C.byref(self._h)))
        self.size = n

    @classmethod
    def _adopt(cls, reg: "RegistrationHIP", handle) -> "DeviceCloud":
        """Wrap an hgs_cloud the library created (hgs_prefilter)."""
        self = cls.__new__(cls)
        self._reg, self._h = reg, handle
        self.size = int(L.lib().hgs_cloud_size(handle))
        return self

    def download(self) -> np.ndarray:
        """The resident cloud as pcl::PointXYZI records (x, y, z, 1, intensity)."""
        from . import synth
        out = np.zeros(self.size, dtype=synth.POINT_XYZI_DTYPE)
        self._reg._check(L.lib().hgs_cloud_download(self._h, out.ctypes.data_as(C.c_void_p), out.dtype.itemsize))
        return out

    @property
    def device_bytes(self) -> int:
        """hgs_cloud_device_bytes: what this keyframe costs while it stays resident (points, index, covariances, voxel tables)."""
        return int(L.lib().hgs_cloud_device_bytes(self._h))

    def invalidate(self):
        L.lib().hgs_cloud_invalidate(self._h)

    def close(self):
        if self._h:
            self._reg._lib.hgs_cloud_destroy(self._h)   # the library that created it
            self._h = C.c_void_p()

    def __del__(self):
        # also after the engine has been closed: hgs_destroy orphans the clouds it created (their device memory goes with the
        # engine) and hgs_cloud_destroy of an orphan only frees the host-side object
        try:
            self.close()
        except Exception:
            pass


class _SearchMethodTarget:
    def __init__(self, reg):
        self._reg = reg

    def nearestKSearch(self, points: np.ndarray, k: int = 1):
        """(indices [n], squared distances [n]) of the exact nearest target point of every query (k must be 1)."""
        if k != 1:
            raise NotImplementedError("the callers only use k = 1 (scan_matching_odometry_nodelet.cpp:316)")
        arr, n, stride = L.cloud_args(points)
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        self._reg._check(L.lib().hgs_nn_target(self._reg._h, arr.ctypes.data_as(C.c_void_p), n, stride, idx.ctypes.data_as(C.c_void_p),
                                               d2.ctypes.data_as(C.c_void_p)))
        return idx, d2


class RegistrationHIP:
    def __init__(self, params: L.HgsParams):
        self.params = params
        self._h = C.c_void_p()
        self._lib = L.lib()
        rc = self._lib.hgs_create(C.byref(params), C.byref(self._h))
        if rc != L.HGS_OK:
            msg = L.lib().hgs_last_error(None)
            raise HgsError(f"hgs_create failed: {L.STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")
        self._result = None
        self._source_n = 0
        self._source_stride = 0
        self._keep = {}
        # measurement / test knobs (A/B scripts, tests that force a code path): HGS_ENGINE_OPTIONS="knn_replay=1,batch_lanes=1" — read HERE, by the
        # test / bench harness; the library itself reads no tuning variable from the environment (include/hgs_registration.h, hgs_debug_set_option)
        for item in filter(None, os.environ.get("HGS_ENGINE_OPTIONS", "").split(",")):
            key, _, value = item.partition("=")
            self.set_option(key.strip(), int(value))

    def set_option(self, key: str, value: int):
        self._check(L.lib().hgs_debug_set_option(self._h, key.encode(), int(value)))

    # ---- life cycle
    def close(self):
        if self._h:
            self._lib.hgs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != L.HGS_OK:
            msg = L.lib().hgs_last_error(self._h)
            raise HgsError(f"{L.STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")

    # ---- pcl::Registration surface
    def setInputTarget(self, cloud):
        if isinstance(cloud, DeviceCloud):
            self._keep["t"] = cloud
            self._check(L.lib().hgs_set_target_cloud(self._h, cloud._h))
        else:
            arr, n, stride = L.cloud_args(cloud)
            self._check(L.lib().hgs_set_target(self._h, arr.ctypes.data_as(C.c_void_p), n, stride))

    def setInputSource(self, cloud):
        if isinstance(cloud, DeviceCloud):
            self._keep["s"] = cloud
            self._source_n = cloud.size
            self._check(L.lib().hgs_set_source_cloud(self._h, cloud._h))
        else:
            arr, n, stride = L.cloud_args(cloud)
            self._source_n = n
            self._check(L.lib().hgs_set_source(self._h, arr.ctypes.data_as(C.c_void_p), n, stride))

    def align(self, guess=None, return_cloud: bool = False):
        """registration->align(*aligned, guess). Returns the hgs_result (and the aligned xyz cloud if requested)."""
        g = L.colmajor16(np.eye(4) if guess is None else guess)
        r = L.HgsResult()
        self._check(L.lib().hgs_align(self._h, L.fptr(g), C.byref(r)))
        self._result = r
        if return_cloud:
            return r, self.transformed_source(r.matrix())
        return r

    def transformed_source(self, T) -> np.ndarray:
        out = np.zeros((self._source_n, 4), np.float32)
        g = L.colmajor16(T)
        self._check(L.lib().hgs_transform_source(self._h, L.fptr(g), out.ctypes.data_as(C.c_void_p), 16))
        return out

    def hasConverged(self) -> bool:
        return bool(self._result.converged) if self._result is not None else False

    def getFinalTransformation(self) -> np.ndarray:
        return self._result.matrix() if self._result is not None else np.eye(4, dtype=np.float32)

    def getFitnessScore(self, max_range: float = L.DBL_MAX, T=None) -> float:
        g = L.colmajor16(self.getFinalTransformation() if T is None else T)
        score, ninl = C.c_double(), C.c_uint32()
        self._check(L.lib().hgs_fitness(self._h, L.fptr(g), float(max_range), C.byref(score), C.byref(ninl)))
        self.last_num_inliers = ninl.value
        return score.value

    def getSearchMethodTarget(self) -> _SearchMethodTarget:
        return _SearchMethodTarget(self)

    # ---- device clouds / batch (LoopDetector::matching)
    def upload(self, cloud) -> DeviceCloud:
        return DeviceCloud(self, cloud)

    def prefilter(self, cloud, params=None, imu_angular_velocity=None, scan_period: float = 0.1) -> DeviceCloud:
        """PrefilteringNodelet::cloud_callback (apps/prefiltering_nodelet.cpp:106-136) on the device: [deskewing ->] distance
        filter -> voxel grid -> outlier removal.  Returns a resident cloud usable as setInputSource / setInputTarget argument.
        imu_angular_velocity: the angular_velocity of the sensor_msgs/Imu sample `select_imu_sample` picks (None = the
        nodelet's empty imu_queue or deskewing off: no deskewing)."""
        if params is None:
            params = L.HgsPrefilterParams()
            self._check(L.lib().hgs_prefilter_params_default(C.byref(params)))
        arr, n, stride = L.cloud_args(cloud)
        h = C.c_void_p()
        if imu_angular_velocity is None:
            self._check(L.lib().hgs_prefilter(self._h, arr.ctypes.data_as(C.c_void_p), n, stride, C.byref(params), C.byref(h)))
        else:
            w = np.ascontiguousarray(imu_angular_velocity, np.float64).reshape(3)
            self._check(L.lib().hgs_prefilter_deskewed(self._h, arr.ctypes.data_as(C.c_void_p), n, stride, C.byref(params), w.ctypes.data_as(C.c_void_p),
                                                       float(scan_period), C.byref(h)))
        return DeviceCloud._adopt(self, h)

    def map_cloud(self, keyframes, poses, resolution: float) -> DeviceCloud:
        """MapCloudGenerator::generate (src/hdl_graph_slam/map_cloud_generator.cpp:13-51) over resident keyframe clouds."""
        n = len(keyframes)
        arr = (C.c_void_p * max(n, 1))(*[k._h for k in keyframes])
        P = np.ascontiguousarray(np.stack([L.colmajor16(p) for p in poses]) if n else np.zeros((0, 16), np.float32))
        h = C.c_void_p()
        self._check(L.lib().hgs_map_cloud_generate(self._h, arr, P.ctypes.data_as(C.c_void_p), n, float(resolution), C.byref(h)))
        return DeviceCloud._adopt(self, h)

    def loop_match_batch(self, candidates, guesses, max_range: float = L.DBL_MAX):
        """Register every candidate DeviceCloud against the current target; returns (records ndarray, best index)."""
        n = len(candidates)
        ptrs = (C.c_void_p * max(n, 1))(*[c._h for c in candidates])
        g = np.ascontiguousarray(np.stack([L.colmajor16(T) for T in guesses]) if n else np.zeros((0, 16), np.float32))
        out = np.zeros(n, dtype=L.RESULT_DTYPE)
        best = C.c_int32(-1)
        self._check(L.lib().hgs_loop_match_batch(self._h, ptrs, n, g.ctypes.data_as(C.c_void_p), float(max_range), out.ctypes.data_as(C.c_void_p),
                                                 C.byref(best)))
        return out, best.value

    # ---- sharded batch: one process per GPU, records all-gathered by RCCL on the engine's stream (include/hgs_registration.h)
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the 128-byte RCCL id every rank passes to comm_init (hand it over with whatever launched the processes)."""
        buf = (C.c_ubyte * 128)()
        rc = L.lib().hgs_comm_get_unique_id(buf)
        if rc != L.HGS_OK:
            raise HgsError(f"hgs_comm_get_unique_id: {L.STATUS.get(rc, rc)}")
        return bytes(buf)

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._check(L.lib().hgs_comm_init(self._h, int(rank), int(world), buf))

    def comm_finalize(self):
        self._check(L.lib().hgs_comm_finalize(self._h))

    def loop_match_batch_sharded(self, candidates, candidate_ids, guesses, n_total: int, max_range: float = L.DBL_MAX, return_status: bool = False):
        """This rank's candidates (DeviceClouds, their positions in the detection's candidate list, their guesses) -> the records of
        ALL n_total candidates in candidate order and the index the sequential rule selects.  Collective over the communicator.
        return_status: do not raise on an error status but return (records, best, status) — a rank whose own share failed has
        still taken part in the exchange and holds the other ranks' records (include/hgs_registration.h)."""
        n = len(candidates)
        ptrs = (C.c_void_p * max(n, 1))(*[c._h for c in candidates])
        ids = np.ascontiguousarray(np.asarray(candidate_ids, np.int32).reshape(-1))
        g = np.ascontiguousarray(np.stack([L.colmajor16(T) for T in guesses]) if n else np.zeros((0, 16), np.float32))
        out = np.zeros(int(n_total), dtype=L.RESULT_DTYPE)
        best = C.c_int32(-1)
        rc = L.lib().hgs_loop_match_batch_sharded(self._h, ptrs, n, ids.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), int(n_total),
                                                  float(max_range), out.ctypes.data_as(C.c_void_p), C.byref(best))
        if return_status:
            return out, best.value, rc
        self._check(rc)
        return out, best.value

    def calc_fitness_score(self, cloud1: DeviceCloud, cloud2: DeviceCloud, relpose, max_range: float = L.DBL_MAX) -> float:
        """InformationMatrixCalculator::calc_fitness_score (src/hdl_graph_slam/information_matrix_calculator.cpp:49-80)."""
        g = L.colmajor16(relpose)
        score = C.c_double()
        self._check(L.lib().hgs_calc_fitness_score(self._h, cloud1._h, cloud2._h, L.fptr(g), float(max_range), C.byref(score)))
        return score.value

    # ---- stage-level hooks (parity tests)
    def nn_target(self, q_xyz):
        return self.getSearchMethodTarget().nearestKSearch(q_xyz, 1)

    def target_covariances(self, n_target: int) -> np.ndarray:
        out = np.zeros((n_target, 6), np.float32)
        self._check(L.lib().hgs_debug_target_covariances(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def gicp_linearize(self, T):
        T12 = np.ascontiguousarray(np.asarray(T, np.float64)[:3, :4])
        H, b, e = np.zeros((6, 6)), np.zeros(6), np.zeros(1)
        corr = np.empty(self._source_n, np.int32)
        vp = C.c_void_p
        self._check(L.lib().hgs_debug_gicp_linearize(self._h, T12.ctypes.data_as(vp), H.ctypes.data_as(vp), b.ctypes.data_as(vp), e.ctypes.data_as(vp),
                                                     corr.ctypes.data_as(vp)))
        return H, b, float(e[0]), corr

    def ndt_cells(self, cap: int = 1 << 20):
        ijk, mean, icov, npts = np.zeros((cap, 3), np.int32), np.zeros((cap, 3)), np.zeros((cap, 6), np.float32), np.zeros(cap, np.int32)
        n = C.c_int32()
        vp = C.c_void_p
        self._check(L.lib().hgs_debug_ndt_cells(self._h, cap, ijk.ctypes.data_as(vp), mean.ctypes.data_as(vp), icov.ctypes.data_as(vp),
                                                npts.ctypes.data_as(vp), C.byref(n)))
        k = min(n.value, cap)
        return ijk[:k].copy(), mean[:k].copy(), icov[:k].copy(), npts[:k].copy()

    def ndt_derivatives(self, p6):
        p = np.ascontiguousarray(p6, np.float64)
        s, g, H = np.zeros(1), np.zeros(6), np.zeros((6, 6))
        vp = C.c_void_p
        self._check(L.lib().hgs_debug_ndt_derivatives(self._h, p.ctypes.data_as(vp), s.ctypes.data_as(vp), g.ctypes.data_as(vp), H.ctypes.data_as(vp)))
        return float(s[0]), g, H

    # ---- measurement
    def profile_enable(self, on: bool = True):
        self._check(L.lib().hgs_profile_enable(self._h, int(on)))

    def profile_read(self, reset: bool = True):
        ms = np.zeros(len(L.STAGES))
        cnt = np.zeros(len(L.STAGES), np.uint64)
        self._check(L.lib().hgs_profile_read(self._h, ms.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), int(reset)))
        return {s: (float(m), int(c)) for s, m, c in zip(L.STAGES, ms, cnt)}

    def synchronize(self):
        self._check(L.lib().hgs_synchronize(self._h))


def select_imu_sample(imu_queue, cloud_stamp):
    """The gyro sample PrefilteringNodelet::deskewing uses (apps/prefiltering_nodelet.cpp:206-216): imu_queue is a list of
    (stamp, angular_velocity) in arrival order; the first sample stamped after the cloud, else the newest one; every sample in
    front of the chosen one is dropped from the queue (the chosen one stays).  Returns the angular velocity, or None for an
    empty queue (the cloud then passes undeskewed)."""
    if not imu_queue:
        return None
    loc = 0
    chosen = imu_queue[0]
    while loc < len(imu_queue):
        chosen = imu_queue[loc]
        if imu_queue[loc][0] > cloud_stamp:
            break
        loc += 1
    del imu_queue[:loc]
    return chosen[1]


def select_best(records: np.ndarray) -> int:
    """The sequential selection rule of loop_detector.hpp:146-153 applied to gathered per-candidate records."""
    rec = np.ascontiguousarray(records)
    best = C.c_int32(-1)
    rc = L.lib().hgs_select_best(rec.ctypes.data_as(C.c_void_p), len(rec), C.byref(best))
    if rc != L.HGS_OK:
        raise HgsError(L.STATUS.get(rc, rc))
    return best.value
