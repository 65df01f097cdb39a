#!/usr/bin/env python3
"""BASELINE config 3 (and 1): a LiDAR odometry stream through ScanMatchingOdometry (mirror of
scan_matching_odometry_nodelet.cpp:165-262) on the HIP backend — per-scan latency (host buffer in, pose out, i.e. including
the H2D upload of every sweep), registrations/s, trajectory error vs ground truth and vs the CPU oracle on the same stream.

  python scripts/odometry_stream.py [--sensor HDL-64E] [--method NDT_OMP] [--scans 40] [--oracle-scans 6]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
from hdl_graph_slam_amd import synth, workloads  # noqa: E402
from hdl_graph_slam_amd.odometry import ScanMatchingOdometry  # noqa: E402
from hdl_graph_slam_amd.registrations import select_registration_method  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sensor", default="HDL-64E")
    ap.add_argument("--method", default="NDT_OMP")
    ap.add_argument("--scans", type=int, default=40)
    ap.add_argument("--oracle-scans", type=int, default=6)
    ap.add_argument("--downsample", type=float, default=0.0)
    ap.add_argument("--ndt-line-search", action="store_true", help="opt-in More-Thuente search (not the reference behaviour)")
    ap.add_argument("--speed", type=float, default=8.0, help="vehicle speed [m/s] at 10 Hz sweeps (8 m/s = KITTI-like 0.8 m per sweep)")
    args = ap.parse_args()
    pnh = {"registration_method": args.method, "reg_resolution": 1.0, "reg_ndt_line_search": args.ndt_line_search}
    stream = workloads.make_odometry_stream(args.sensor, 0, args.scans, speed=args.speed, downsample=args.downsample or None)
    # keyframe rule of launch/hdl_graph_slam_kitti.launch:41-43
    kf = dict(keyframe_delta_trans=5.0, keyframe_delta_angle=2.0, keyframe_delta_time=10000.0)
    reg = select_registration_method(pnh, device_id=0)
    od = ScanMatchingOdometry(reg, **kf)
    lat, est, its = [], [], []
    for t, c in zip(stream.stamps, stream.scans):
        t0 = time.perf_counter()
        est.append(od.matching(t, c))
        lat.append(time.perf_counter() - t0)
        its.append(od.last_result.iterations if od.last_result is not None else 0)
    gt0 = np.linalg.inv(stream.poses[0])
    err = [synth.pose_error(e, gt0 @ p) for e, p in zip(est, stream.poses)]
    lat_ms = np.array(lat[2:]) * 1e3      # skip the keyframe-only first call and the first (warm-up) registration
    out = {"sensor": args.sensor, "method": args.method, "speed_mps": args.speed, "points_per_scan": int(np.mean([len(c) for c in stream.scans])), "scans": args.scans,
           "latency_ms": {"p50": round(float(np.percentile(lat_ms, 50)), 3), "p90": round(float(np.percentile(lat_ms, 90)), 3),
                          "p99": round(float(np.percentile(lat_ms, 99)), 3), "max": round(float(lat_ms.max()), 3)},
           "registrations_per_sec": round(float(len(lat_ms) / (lat_ms.sum() * 1e-3)), 2), "mean_iterations": float(np.mean(its[1:])),
           "keyframes": od.num_keyframes,
           "trajectory_error_vs_gt": {"final_translation_m": round(err[-1][0], 4), "rmse_translation_m": round(float(np.sqrt(np.mean([e[0] ** 2 for e in err]))), 4),
                                      "rmse_rotation_rad": round(float(np.sqrt(np.mean([e[1] ** 2 for e in err]))), 5)}}
    if args.oracle_scans > 0:
        import oracle as O
        p = O.HgsParams()
        for name, _ in O.HgsParams._fields_:
            setattr(p, name, getattr(reg.params, name))
        O.set_num_threads(min(32, os.cpu_count() or 1))
        oo = ScanMatchingOdometry(O.OracleRegistration(p), **kf)
        k = min(args.oracle_scans, args.scans)
        t0 = time.perf_counter()
        ref = [oo.matching(t, c) for t, c in zip(stream.stamps[:k], stream.scans[:k])]
        cpu = time.perf_counter() - t0
        d = [synth.pose_error(a, b) for a, b in zip(est[:k], ref)]
        out["vs_cpu_oracle"] = {"scans": k, "max_translation_diff_m": float(max(x[0] for x in d)), "max_rotation_diff_rad": float(max(x[1] for x in d)),
                                "cpu_ms_per_scan": round(cpu / max(k - 1, 1) * 1e3, 1), "threads": O.set_num_threads(0)}
    print(json.dumps(out))
    reg.close()


if __name__ == "__main__":
    main()
