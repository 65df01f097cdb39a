#!/usr/bin/env python3
"""Generates tests/golden/vlp16_next_rows_v3.npz — known-answer vectors for the round-2 additions to the "next" rows: the
prefilter's deskewing step and pcl::ApproximateVoxelGrid (row f2 / a3), and the map cloud in pcl::octree traversal order (row
f3).  Same rationale as make_golden.py: the reference has no golden vectors; these freeze the oracle's restatement (which
tests/test_prefilter.py and tests/test_map_cloud.py pin against independent numpy / pure-Python statements) and give the HIP
path fixed inputs and outputs.  Re-run: python tests/golden/make_golden_v3.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402

IMU_W, SCAN_PERIOD = [0.31, -0.17, 0.93], 0.1
MAP_RES = 0.4


def inputs():
    scene = synth.make_scene(3)
    raw = synth.scan(scene, "VLP-16", synth.pose_matrix([0, 0, 0], [0, 0, 0]), 903)[::4]
    rng = np.random.default_rng(3)
    raw["intensity"] = rng.uniform(0, 255, len(raw)).astype(np.float32)
    bad = synth.to_xyzi(np.array([[np.nan, 0, 0]], np.float32))
    raw = np.concatenate([raw[:500], bad, raw[500:]])
    kf = [synth.scan(scene, "VLP-16", synth.pose_matrix([3.0 * k, 0.5 * k, 0], [0, 0, 0.1 * k]), 950 + k)[::6] for k in range(3)]
    poses = [synth.pose_matrix([3.0 * k, 0.5 * k, 0], [0, 0, 0.1 * k]).astype(np.float32) for k in range(3)]
    return raw, kf, poses


def cases():
    p = O.default_prefilter_params()                       # distance 1..100 m, VOXELGRID 0.1, STATISTICAL 20 / 1.0
    yield "deskew_default", p, IMU_W
    q = O.default_prefilter_params()
    q.downsample_method, q.downsample_resolution, q.outlier_removal_method = 2, 0.3, 0
    yield "approx_voxelgrid_0.3", q, None
    r = O.default_prefilter_params()
    r.downsample_method, r.downsample_resolution, r.outlier_removal_method, r.radius_radius, r.radius_min_neighbors = 2, 0.5, 2, 1.0, 2
    yield "deskew_approx_radius", r, IMU_W


def main():
    raw, kf, poses = inputs()
    out = {"raw_xyzi": np.stack([raw["x"], raw["y"], raw["z"], raw["intensity"]], axis=1)}
    for name, p, w in cases():
        out[name] = O.prefilter(raw, p, imu_angular_velocity=w, scan_period=SCAN_PERIOD)
    for k, c in enumerate(kf):
        out[f"kf{k}_xyzi"] = np.stack([c["x"], c["y"], c["z"], c["intensity"]], axis=1)
    out["poses"] = np.stack(poses)
    out["map_cloud"] = O.map_cloud(kf, poses, MAP_RES)
    path = os.path.join(HERE, "vlp16_next_rows_v3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
