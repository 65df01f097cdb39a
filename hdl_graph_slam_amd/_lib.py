"""ctypes binding of libhgs_hip.so (the C-ABI of include/hgs_registration.h).

The HIP extension is the product: if it is missing or cannot be loaded this module raises — there is no CPU
fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhgs_hip.so")

HGS_OK = 0
HGS_ERR_INVALID_ARGUMENT, HGS_ERR_NO_TARGET, HGS_ERR_NO_SOURCE, HGS_ERR_HIP, HGS_ERR_NO_DEVICE, HGS_ERR_UNSUPPORTED, HGS_ERR_OUT_OF_MEMORY, HGS_ERR_INTERNAL, HGS_ERR_COMM = range(1, 10)
HGS_FAST_GICP, HGS_FAST_VGICP, HGS_NDT_OMP = 0, 1, 2
HGS_KDTREE, HGS_DIRECT1, HGS_DIRECT7, HGS_DIRECT27 = 0, 1, 2, 3
HGS_REG_FROBENIUS, HGS_REG_PLANE, HGS_REG_MIN_EIG, HGS_REG_NORMALIZED_MIN_EIG, HGS_REG_NONE = 0, 1, 2, 3, 4
STAGES = ["upload", "index", "covariance", "voxelize", "linearize", "error", "solve", "fitness", "prefilter"]
DBL_MAX = float(np.finfo(np.float64).max)

STATUS = {1: "invalid argument", 2: "no target set", 3: "no source set", 4: "HIP runtime error", 5: "no usable HIP device", 6: "unsupported",
          7: "out of host memory inside the backend", 8: "internal error (C++ exception caught at the C boundary)", 9: "communicator error (RCCL)"}


class HgsParams(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("max_iterations", C.c_int32),
        ("transformation_epsilon", C.c_double), ("rotation_epsilon", C.c_double),
        ("max_correspondence_distance", C.c_double),
        ("correspondence_randomness", C.c_int32), ("neighbor_search", C.c_int32),
        ("resolution", C.c_double), ("ndt_step_size", C.c_double), ("ndt_outlier_ratio", C.c_double),
        ("ndt_min_points_per_voxel", C.c_int32), ("ndt_upstream_hd1_sign", C.c_int32),
        ("lm_max_iterations", C.c_int32), ("lm_init_lambda_factor", C.c_double),
        ("device_id", C.c_int32), ("regularization_method", C.c_int32),
        ("ndt_line_search", C.c_int32), ("reserved", C.c_int32),
    ]


class HgsPrefilterParams(C.Structure):
    """hgs_prefilter_params: the rosparams of apps/prefiltering_nodelet.cpp:51-96."""
    _fields_ = [
        ("use_distance_filter", C.c_int32), ("downsample_method", C.c_int32),
        ("distance_near_thresh", C.c_double), ("distance_far_thresh", C.c_double), ("downsample_resolution", C.c_double),
        ("outlier_removal_method", C.c_int32), ("statistical_mean_k", C.c_int32),
        ("statistical_stddev", C.c_double), ("radius_radius", C.c_double),
        ("radius_min_neighbors", C.c_int32), ("reserved", C.c_int32),
    ]


HGS_DOWNSAMPLE_NONE, HGS_DOWNSAMPLE_VOXELGRID, HGS_DOWNSAMPLE_APPROX_VOXELGRID = 0, 1, 2
HGS_OUTLIER_NONE, HGS_OUTLIER_STATISTICAL, HGS_OUTLIER_RADIUS = 0, 1, 2


class HgsResult(C.Structure):
    _fields_ = [
        ("final_transformation", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32),
        ("error", C.c_double), ("fitness_score", C.c_double), ("num_inliers", C.c_uint32),
        ("candidate_id", C.c_int32), ("lm_tries", C.c_int32), ("reserved", C.c_int32),
    ]

    def matrix(self) -> np.ndarray:
        return np.array(self.final_transformation, dtype=np.float32).reshape(4, 4).T.copy()


RESULT_DTYPE = np.dtype([("final_transformation", "<f4", (16,)), ("converged", "<i4"), ("iterations", "<i4"), ("error", "<f8"),
                         ("fitness_score", "<f8"), ("num_inliers", "<u4"), ("candidate_id", "<i4"), ("lm_tries", "<i4"), ("reserved", "<i4")])
assert RESULT_DTYPE.itemsize == C.sizeof(HgsResult)

# every symbol include/hgs_registration.h declares
EXPORTS = [
    "hgs_params_default", "hgs_create", "hgs_destroy", "hgs_last_error", "hgs_abi_version",
    "hgs_cloud_create", "hgs_cloud_destroy", "hgs_cloud_size", "hgs_cloud_device_bytes", "hgs_cloud_invalidate",
    "hgs_set_target", "hgs_set_target_cloud", "hgs_set_source", "hgs_set_source_cloud",
    "hgs_align", "hgs_transform_source", "hgs_fitness", "hgs_nn_target",
    "hgs_loop_match_batch", "hgs_select_best", "hgs_calc_fitness_score",
    "hgs_comm_get_unique_id", "hgs_comm_init", "hgs_comm_finalize", "hgs_loop_match_batch_sharded",
    "hgs_prefilter_params_default", "hgs_prefilter", "hgs_prefilter_deskewed", "hgs_cloud_download", "hgs_map_cloud_generate",
    "hgs_profile_enable", "hgs_profile_read", "hgs_synchronize",
    "hgs_debug_target_covariances", "hgs_debug_gicp_linearize", "hgs_debug_ndt_cells", "hgs_debug_ndt_derivatives", "hgs_debug_merge_shard_records", "hgs_debug_set_option",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build the MI355X backend first (python -c 'import __graft_entry__ as g; g.build()'). "
                           "hdl_graph_slam_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, fp = C.c_void_p, C.c_size_t, C.POINTER(C.c_float)
    L.hgs_params_default.argtypes = [C.c_int32, C.POINTER(HgsParams)]
    L.hgs_create.argtypes = [C.POINTER(HgsParams), C.POINTER(vp)]
    L.hgs_destroy.argtypes = [vp]
    L.hgs_last_error.argtypes = [vp]
    L.hgs_last_error.restype = C.c_char_p
    L.hgs_cloud_create.argtypes = [vp, vp, sz, sz, C.POINTER(vp)]
    L.hgs_cloud_destroy.argtypes = [vp]
    L.hgs_cloud_size.argtypes = [vp]
    L.hgs_cloud_size.restype = sz
    L.hgs_cloud_device_bytes.argtypes = [vp]
    L.hgs_cloud_device_bytes.restype = sz
    L.hgs_cloud_invalidate.argtypes = [vp]
    L.hgs_set_target.argtypes = [vp, vp, sz, sz]
    L.hgs_set_source.argtypes = [vp, vp, sz, sz]
    L.hgs_set_target_cloud.argtypes = [vp, vp]
    L.hgs_set_source_cloud.argtypes = [vp, vp]
    L.hgs_align.argtypes = [vp, fp, C.POINTER(HgsResult)]
    L.hgs_transform_source.argtypes = [vp, fp, vp, sz]
    L.hgs_fitness.argtypes = [vp, fp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.hgs_nn_target.argtypes = [vp, vp, sz, sz, vp, vp]
    L.hgs_loop_match_batch.argtypes = [vp, C.POINTER(vp), sz, vp, C.c_double, vp, C.POINTER(C.c_int32)]
    L.hgs_select_best.argtypes = [vp, sz, C.POINTER(C.c_int32)]
    L.hgs_comm_get_unique_id.argtypes = [vp]
    L.hgs_comm_init.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.hgs_comm_finalize.argtypes = [vp]
    L.hgs_loop_match_batch_sharded.argtypes = [vp, C.POINTER(vp), sz, vp, vp, sz, C.c_double, vp, C.POINTER(C.c_int32)]
    L.hgs_calc_fitness_score.argtypes = [vp, vp, vp, fp, C.c_double, C.POINTER(C.c_double)]
    L.hgs_prefilter_params_default.argtypes = [C.POINTER(HgsPrefilterParams)]
    L.hgs_prefilter.argtypes = [vp, vp, sz, sz, C.POINTER(HgsPrefilterParams), C.POINTER(vp)]
    L.hgs_prefilter_deskewed.argtypes = [vp, vp, sz, sz, C.POINTER(HgsPrefilterParams), vp, C.c_double, C.POINTER(vp)]
    L.hgs_cloud_download.argtypes = [vp, vp, sz]
    L.hgs_map_cloud_generate.argtypes = [vp, C.POINTER(vp), vp, sz, C.c_double, C.POINTER(vp)]
    L.hgs_profile_enable.argtypes = [vp, C.c_int]
    L.hgs_profile_read.argtypes = [vp, vp, vp, C.c_int]
    L.hgs_synchronize.argtypes = [vp]
    L.hgs_debug_target_covariances.argtypes = [vp, vp]
    L.hgs_debug_set_option.argtypes = [vp, C.c_char_p, C.c_int32]
    L.hgs_debug_gicp_linearize.argtypes = [vp, vp, vp, vp, vp, vp]
    L.hgs_debug_ndt_cells.argtypes = [vp, C.c_int32, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    L.hgs_debug_ndt_derivatives.argtypes = [vp, vp, vp, vp, vp]
    L.hgs_debug_merge_shard_records.argtypes = [vp, vp, C.c_int32, C.c_size_t, C.c_size_t, vp, vp]
    _lib = L
    return L


def default_params(method: int) -> HgsParams:
    p = HgsParams()
    rc = lib().hgs_params_default(method, C.byref(p))
    if rc != HGS_OK:
        raise ValueError(f"hgs_params_default({method}) -> {STATUS.get(rc, rc)}")
    return p


def cloud_args(cloud: np.ndarray):
    """(contiguous array, n, stride) for PointXYZI records or [n,>=3] float32 arrays."""
    cloud = np.ascontiguousarray(cloud)
    if cloud.dtype.fields is not None:
        return cloud, len(cloud), cloud.dtype.itemsize
    cloud = np.ascontiguousarray(cloud, dtype=np.float32)
    if cloud.ndim != 2 or cloud.shape[1] < 3:
        raise ValueError("expected [n, >=3] float32 or PointXYZI records")
    return cloud, cloud.shape[0], cloud.shape[1] * 4  # (numpy reports stride 0 for empty arrays)


def colmajor16(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T.reshape(16))


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))
