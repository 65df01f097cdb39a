"""select_registration_method — Python mirror of include/hdl_graph_slam/registrations.hpp:17 /
src/hdl_graph_slam/registrations.cpp:22-124 for the engines this backend implements.

`pnh` is any mapping of the nodelet's private rosparams (the same keys and defaults the reference reads with
pnh.param<T>(name, default)).  Returns a RegistrationHIP whose methods mirror pcl::Registration."""
from __future__ import annotations

import sys

from . import _lib as L
from .registration import RegistrationHIP

# Engines the reference factory builds from third-party CPU libraries and that are NOT rebuilt here.  The
# reference routes by substring (registrations.cpp:57-99): "ICP" exactly -> pcl::IterativeClosestPoint; any other name containing
# "GICP" -> pcl / pclomp GeneralizedIterativeClosestPoint; everything else -> NDT, pclomp's if the name contains "OMP", PCL's
# single-threaded pcl::NormalDistributionsTransform (another algorithm: KDTREE neighbourhoods, its own line search) otherwise.


_REGULARIZATION = {"FROBENIUS": L.HGS_REG_FROBENIUS, "PLANE": L.HGS_REG_PLANE, "MIN_EIG": L.HGS_REG_MIN_EIG,
                   "NORMALIZED_MIN_EIG": L.HGS_REG_NORMALIZED_MIN_EIG, "NONE": L.HGS_REG_NONE}


def _regularization(pnh) -> int:
    """Not a rosparam of the reference: registrations.cpp never calls fast_gicp's setRegularizationMethod, so the fast_gicp
    constructor default applies (FROBENIUS).  `reg_regularization_method` lets a deployment whose fast_gicp
    checkout defaults to another method (PLANE) match it without recompiling."""
    name = str(pnh.get("reg_regularization_method", "FROBENIUS")).upper()
    if name not in _REGULARIZATION:
        raise ValueError(f"reg_regularization_method={name}: expected one of {sorted(_REGULARIZATION)}")
    return _REGULARIZATION[name]


def params_from_rosparams(pnh) -> L.HgsParams:
    get = pnh.get
    method = str(get("registration_method", "NDT_OMP"))                            # registrations.cpp:26
    if method in ("FAST_GICP", "FAST_GICP_HIP"):                                   # registrations.cpp:27-36
        p = L.default_params(L.HGS_FAST_GICP)
        p.transformation_epsilon = float(get("reg_transformation_epsilon", 0.01))
        p.max_iterations = int(get("reg_maximum_iterations", 64))
        p.max_correspondence_distance = float(get("reg_max_correspondence_distance", 2.5))
        p.correspondence_randomness = int(get("reg_correspondence_randomness", 20))
        p.regularization_method = _regularization(pnh)
        return p
    if method in ("FAST_VGICP", "FAST_VGICP_CUDA", "FAST_VGICP_HIP"):              # registrations.cpp:37-56
        p = L.default_params(L.HGS_FAST_VGICP)
        p.resolution = float(get("reg_resolution", 1.0))
        p.transformation_epsilon = float(get("reg_transformation_epsilon", 0.01))
        p.max_iterations = int(get("reg_maximum_iterations", 64))
        p.correspondence_randomness = int(get("reg_correspondence_randomness", 20))
        p.regularization_method = _regularization(pnh)
        return p
    cpu_engine = None
    if method == "ICP":                                                             # registrations.cpp:57-64
        cpu_engine = "pcl::IterativeClosestPoint"
    elif "GICP" in method:                                                          # registrations.cpp:65-87 (GICP, GICP_OMP, GICP_FOO, ...)
        cpu_engine = "pclomp::GeneralizedIterativeClosestPoint" if "OMP" in method else "pcl::GeneralizedIterativeClosestPoint"
    elif "OMP" not in method and method != "NDT_HIP":                               # registrations.cpp:94-100 (NDT, MY_NDT, typos, ...)
        cpu_engine = "pcl::NormalDistributionsTransform"
    if cpu_engine is not None:
        raise NotImplementedError(f"registration_method={method}: the reference's factory builds {cpu_engine} for this name, a CPU engine this backend "
                                  "does not replace (it implements FAST_GICP, FAST_VGICP and NDT_OMP); running the device NDT_OMP instead would change the result silently")
    if "NDT" not in method:                                                         # registrations.cpp:88-91 (then "... _OMP": pclomp's NDT)
        print(f"warning: unknown registration type({method})\n       : use NDT", file=sys.stderr)
    p = L.default_params(L.HGS_NDT_OMP)                                             # registrations.cpp:93,101-120
    p.resolution = float(get("reg_resolution", 0.5))
    p.transformation_epsilon = float(get("reg_transformation_epsilon", 0.01))
    p.max_iterations = int(get("reg_maximum_iterations", 64))
    nn = str(get("reg_nn_search_method", "DIRECT7"))
    p.neighbor_search = {"KDTREE": L.HGS_KDTREE, "DIRECT1": L.HGS_DIRECT1}.get(nn, L.HGS_DIRECT7)      # registrations.cpp:112-118
    # Not a rosparam of the reference: ndt_omp's More-Thuente loop never executes, which is what 0 reproduces;
    # 1 turns on a working line search (fewer, better-behaved iterations; NOT the reference's result).
    p.ndt_line_search = 1 if get("reg_ndt_line_search", False) in (True, 1, "1", "true", "True") else 0
    return p


def select_registration_method(pnh, device_id: int = 0) -> RegistrationHIP:
    p = params_from_rosparams(pnh)
    p.device_id = device_id
    # reg_num_threads (registrations.cpp:30,51,102) has no meaning on the device: parallelism is over points and candidates
    return RegistrationHIP(p)
