"""Where a sweep of the KITTI launch file's pipeline (device prefilter -> FAST_GICP, bench.py --config 3 `kitti_launch_fast_gicp`) spends its host-visible time:
hgs_prefilter, hgs_set_source_cloud, hgs_align, per call, p50 over the sweeps of one stream."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdl_graph_slam_amd import workloads, _lib as L  # noqa: E402
from hdl_graph_slam_amd.registrations import select_registration_method  # noqa: E402

stream = workloads.make_odometry_stream("HDL-64E", 0, 32, speed=8.0)
reg = select_registration_method({"registration_method": "FAST_GICP", "reg_transformation_epsilon": 0.1, "reg_max_correspondence_distance": 2.0}, device_id=0)
pp = L.HgsPrefilterParams()
L.lib().hgs_prefilter_params_default(C.byref(pp))
pp.use_distance_filter, pp.distance_near_thresh, pp.distance_far_thresh = 1, 0.1, 100.0
pp.downsample_method, pp.downsample_resolution = L.HGS_DOWNSAMPLE_VOXELGRID, 0.25
pp.outlier_removal_method, pp.radius_radius, pp.radius_min_neighbors = L.HGS_OUTLIER_RADIUS, 0.5, 2
kf = reg.prefilter(stream.scans[0], pp)
reg.setInputTarget(kf)
t_pf, t_set, t_al, its = [], [], [], []
prev = np.eye(4)
for c in stream.scans[1:]:
    t0 = time.perf_counter()
    d = reg.prefilter(c, pp)
    t1 = time.perf_counter()
    reg.setInputSource(d)
    t2 = time.perf_counter()
    r = reg.align(prev)
    t3 = time.perf_counter()
    prev = r.matrix()
    t_pf.append((t1 - t0) * 1e3), t_set.append((t2 - t1) * 1e3), t_al.append((t3 - t2) * 1e3), its.append(r.iterations)
med = lambda v: round(float(np.median(v[3:])), 4)
print("points after prefilter", d.size, "| hgs_prefilter p50 ms", med(t_pf), "| set_source_cloud", med(t_set), "| hgs_align", med(t_al), "| iterations p50", float(np.median(its)))
