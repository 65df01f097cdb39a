"""ScanMatchingOdometry — Python mirror of ScanMatchingOdometryNodelet::matching
(apps/scan_matching_odometry_nodelet.cpp:165-262), the odometry caller of the registration engine: frame-to-keyframe
matching with the previous result as the initial guess and a keyframe switch on translation / rotation / time deltas.

The registration object is anything with the pcl::Registration surface the nodelet uses (setInputTarget,
setInputSource, align, hasConverged, getFinalTransformation): hdl_graph_slam_amd.RegistrationHIP on an MI355X, or the CPU
oracle inside the tests.  ROS plumbing (TF, IMU / robot-odometry guesses, status publishing) is out of scope; the
msf_delta hook is kept as an optional argument because it multiplies into the guess (:210).
"""
from __future__ import annotations

import numpy as np


def make_downsample(registration, downsample_method: str = "VOXELGRID", downsample_resolution: float = 0.1):
    """The downsample filter of the odometry nodelet (scan_matching_odometry_nodelet.cpp:84-104, applied at :147-157).

    VOXELGRID (pcl::VoxelGrid) and APPROX_VOXELGRID (pcl::ApproximateVoxelGrid, :91-96) run on the device through hgs_prefilter
    (distance filter and outlier removal off) and return a RESIDENT cloud, so the sweep is uploaded once and registered in
    place; NONE is the pass-through filter the launch files select for this nodelet."""
    method = downsample_method.upper()
    if method in ("VOXELGRID", "APPROX_VOXELGRID"):
        from . import _lib as L
        import ctypes as C
        p = L.HgsPrefilterParams()
        L.lib().hgs_prefilter_params_default(C.byref(p))
        p.use_distance_filter, p.outlier_removal_method = 0, L.HGS_OUTLIER_NONE
        p.downsample_method = L.HGS_DOWNSAMPLE_VOXELGRID if method == "VOXELGRID" else L.HGS_DOWNSAMPLE_APPROX_VOXELGRID
        p.downsample_resolution = downsample_resolution
        return lambda cloud: registration.prefilter(cloud, p)
    return lambda cloud: cloud   # NONE / unknown -> passthrough (:97-103)


def _rotation_angle(R: np.ndarray) -> float:
    """acos(Eigen::Quaternionf(R).w()) — HALF the rotation angle, exactly what the reference thresholds (:226,243)."""
    tr = float(R[0, 0] + R[1, 1] + R[2, 2])
    w = 0.5 * np.sqrt(max(tr + 1.0, 0.0))
    return float(np.arccos(min(1.0, max(-1.0, w))))


class ScanMatchingOdometry:
    def __init__(self, registration, keyframe_delta_trans: float = 0.25, keyframe_delta_angle: float = 0.15, keyframe_delta_time: float = 1.0,
                 transform_thresholding: bool = False, max_acceptable_trans: float = 1.0, max_acceptable_angle: float = 1.0, downsample=None):
        # defaults: scan_matching_odometry_nodelet.cpp:76-83
        self.registration = registration
        self.keyframe_delta_trans = keyframe_delta_trans
        self.keyframe_delta_angle = keyframe_delta_angle
        self.keyframe_delta_time = keyframe_delta_time
        self.transform_thresholding = transform_thresholding
        self.max_acceptable_trans = max_acceptable_trans
        self.max_acceptable_angle = max_acceptable_angle
        self.downsample = downsample or (lambda cloud: cloud)   # :147-157 (launch files use NONE for this nodelet)
        self.keyframe = None
        self.keyframe_pose = np.eye(4, dtype=np.float32)
        self.keyframe_stamp = 0.0
        self.prev_trans = np.eye(4, dtype=np.float32)
        self.num_keyframes = 0
        self.last_result = None

    def matching(self, stamp: float, cloud, msf_delta=None) -> np.ndarray:
        """Returns odom = keyframe_pose * trans (Eigen::Matrix4f semantics: float32 products)."""
        if self.keyframe is None:                                   # :166-174
            self.prev_trans = np.eye(4, dtype=np.float32)
            self.keyframe_pose = np.eye(4, dtype=np.float32)
            self.keyframe_stamp = stamp
            self.keyframe = self.downsample(cloud)
            self.registration.setInputTarget(self.keyframe)
            self.num_keyframes = 1
            return np.eye(4, dtype=np.float32)
        filtered = self.downsample(cloud)                           # :176-177
        self.registration.setInputSource(filtered)
        guess = self.prev_trans if msf_delta is None else (self.prev_trans @ np.asarray(msf_delta, np.float32)).astype(np.float32)
        result = self.registration.align(guess)                     # :210
        self.last_result = result
        if not result.converged:                                    # :214-218
            return (self.keyframe_pose @ self.prev_trans).astype(np.float32)
        trans = np.asarray(result.matrix(), np.float32)             # :220
        odom = (self.keyframe_pose @ trans).astype(np.float32)
        if self.transform_thresholding:                             # :223-233
            delta = (np.linalg.inv(self.prev_trans) @ trans).astype(np.float32)
            dx = float(np.linalg.norm(delta[:3, 3]))
            da = _rotation_angle(delta[:3, :3])
            if dx > self.max_acceptable_trans or da > self.max_acceptable_angle:
                return (self.keyframe_pose @ self.prev_trans).astype(np.float32)
        self.prev_trans = trans                                     # :236
        delta_trans = float(np.linalg.norm(trans[:3, 3]))           # :241-243
        delta_angle = _rotation_angle(trans[:3, :3])
        delta_time = stamp - self.keyframe_stamp
        if delta_trans > self.keyframe_delta_trans or delta_angle > self.keyframe_delta_angle or delta_time > self.keyframe_delta_time:
            self.keyframe = filtered                                # :245-252
            self.registration.setInputTarget(self.keyframe)
            self.keyframe_pose = odom
            self.keyframe_stamp = stamp
            self.prev_trans = np.eye(4, dtype=np.float32)
            self.num_keyframes += 1
        return odom
