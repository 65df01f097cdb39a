"""Per-wave phase clocks and step counts of k_knn_cov on the source cloud of one sweep of the KITTI launch file's pipeline and on config 2's pair
(measurement build: scripts/build_variant.sh knnprobe -DHGS_KNN_PROBE, copied over hdl_graph_slam_amd/lib/libhgs_hip.so by the calling script)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdl_graph_slam_amd import workloads, _lib as L  # noqa: E402
from hdl_graph_slam_amd.registrations import select_registration_method  # noqa: E402

lib = L.lib()
lib.hgs_debug_read_knn_probe.argtypes = [C.c_void_p, C.c_size_t]
lib.hgs_debug_read_knn_probe.restype = C.c_int
N = 1 << 16


def read():
    buf = np.zeros((N, 8), dtype=np.uint64)
    rc = lib.hgs_debug_read_knn_probe(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, rc
    return buf


def report(tag, buf):
    w = buf[buf[:, 0] > 0]
    if len(w) == 0:
        print(tag, "no waves recorded")
        return
    t0 = w[:, 0].astype(np.int64)
    origin = t0.min()
    us = lambda v: (v.astype(np.int64) - origin) / 100.0  # wall_clock64: 100 MHz
    start, pre, walk, end = us(w[:, 0]), us(w[:, 1]), us(w[:, 2]), us(w[:, 3])
    q = lambda v: "p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(v, [10, 50, 90, 100]))
    print(tag, "waves", len(w), "| kernel span us %.1f" % (end.max()))
    print("   start offset    ", q(start))
    print("   prefill us      ", q(pre - start))
    print("   pass-1 walk us  ", q(walk - pre))
    print("   pass-2 gather us", q(end - walk))
    print("   wave total us   ", q(end - start))
    print("   group steps     ", q(w[:, 5].astype(float)), "| leaf visits", q(w[:, 6].astype(float)), "| insertion events", q(w[:, 7].astype(float)))
    tot = end - start
    worst = np.argsort(-tot)[:5]
    for i in worst:
        print("   slow wave: total %.1f prefill %.1f walk %.1f gather %.1f steps %d leaves %d events %d" % (tot[i], (pre - start)[i], (walk - pre)[i], (end - walk)[i], w[i, 5], w[i, 6], w[i, 7]))
    r = np.corrcoef(w[:, 5].astype(float), (walk - pre))[0, 1]
    print("   us per group step (walk / steps) p50 %.2f | corr(steps, walk us) %.2f" % (np.median((walk - pre) / np.maximum(1, w[:, 5].astype(float))), r))


# (a) KITTI pipeline source
stream = workloads.make_odometry_stream("HDL-64E", 0, 6, speed=8.0)
reg = select_registration_method({"registration_method": "FAST_GICP", "reg_transformation_epsilon": 0.1, "reg_max_correspondence_distance": 2.0}, device_id=0)
pp = L.HgsPrefilterParams()
lib.hgs_prefilter_params_default(C.byref(pp))
pp.use_distance_filter, pp.distance_near_thresh, pp.distance_far_thresh = 1, 0.1, 100.0
pp.downsample_method, pp.downsample_resolution = L.HGS_DOWNSAMPLE_VOXELGRID, 0.25
pp.outlier_removal_method, pp.radius_radius, pp.radius_min_neighbors = L.HGS_OUTLIER_RADIUS, 0.5, 2
kf = reg.prefilter(stream.scans[0], pp)
reg.setInputTarget(kf)
prev = np.eye(4)
for k, c in enumerate(stream.scans[1:]):
    d = reg.prefilter(c, pp)
    reg.setInputSource(d)
    read()  # clear
    r = reg.align(prev)
    prev = r.matrix()
    if k >= 2:
        report("kitti source (%d points), sweep %d:" % (d.size, k + 1), read())

# (b) config 2: HDL-32E pair (~65 k points each), both clouds' covariances in ONE launch (blockIdx.y = cloud)
from hdl_graph_slam_amd import synth  # noqa: E402
tgt, src, T = synth.make_pair("HDL-32E", 0)
reg2 = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
d_tgt, d_src = reg2.upload(tgt), reg2.upload(src)
for rep in range(3):
    d_tgt.invalidate(), d_src.invalidate()
    reg2.setInputTarget(d_tgt)
    reg2.setInputSource(d_src)
    read()
    reg2.align(np.eye(4))
    if rep == 2:
        report("config 2 pair (%d + %d points):" % (len(tgt), len(src)), read())
