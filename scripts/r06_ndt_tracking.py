#!/usr/bin/env python3
"""Why does NDT_OMP odometry drift on the synthetic config-3 stream?    CPU only: the ORACLE's stream through
the mirror of ScanMatchingOdometryNodelet::matching (hdl_graph_slam_amd/odometry.py) — the device reproduces the oracle stream bit for bit
(tests/test_odometry.py), so what is learned here holds for the HIP path.

Varied, one at a time, against the round-5 configuration (corridor scene, raw 119 k-point sweeps, NDT_OMP DIRECT7 1.0 m, 3 m/s):
  scene      "corridor": synth.make_scene (boxes kept >= 6 m from the driving line);  "uniform": boxes uniform in xy as SURVEY §8d words it
             (only boxes that would swallow the vehicle are dropped: nothing closer than 1.5 m to the line)
  prefilter  the KITTI launch file's prefiltering nodelet (launch/hdl_graph_slam_kitti.launch:22-34: 0.1..100 m, VoxelGrid 0.25, radius 0.5 / 2)
  line search  the opt-in More-Thuente search (reg_ndt_line_search; ndt_omp itself never runs its loop)
  resolution 1.0 (the launch files) / 2.0
  engine     FAST_GICP with the KITTI launch file's parameters, for scale
Output: a markdown table (profiles/r06_ndt_tracking.md).

  python scripts/r06_ndt_tracking.py [--scans 60] [--out profiles/r06_ndt_tracking.md]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402
from hdl_graph_slam_amd.odometry import ScanMatchingOdometry  # noqa: E402


def uniform_scene(seed, n_boxes=80, n_cyl=20, extent=60.0, clear=1.5):
    """synth.make_scene without the 6 m corridor: centres uniform in xy; a box / cylinder whose footprint comes closer than `clear` to the line y = 0 is dropped."""
    rng = np.random.default_rng(10_000 + seed)
    ctr = rng.uniform(-extent, extent, size=(n_boxes, 2))
    half = np.stack([rng.uniform(0.5, 6.0, n_boxes), rng.uniform(0.5, 6.0, n_boxes)], axis=1)
    wall = rng.random(n_boxes) < 0.33
    axis = rng.integers(0, 2, n_boxes)
    half[wall, 0] = np.where(axis[wall] == 0, rng.uniform(4.0, 15.0, wall.sum()), 0.15)
    half[wall, 1] = np.where(axis[wall] == 0, 0.15, rng.uniform(4.0, 15.0, wall.sum()))
    height = rng.uniform(1.0, 8.0, n_boxes)
    keep = np.abs(ctr[:, 1]) - half[:, 1] > clear
    box_min = np.concatenate([ctr - half, np.zeros((n_boxes, 1))], axis=1)[keep]
    box_max = np.concatenate([ctr + half, height[:, None]], axis=1)[keep]
    cyl_xy = rng.uniform(-extent, extent, size=(n_cyl, 2))
    cyl_r = rng.uniform(0.15, 0.6, n_cyl)
    cyl_h = rng.uniform(3.0, 10.0, n_cyl)
    ck = np.abs(cyl_xy[:, 1]) - cyl_r > clear
    return synth.Scene(box_min.astype(np.float64), box_max.astype(np.float64), cyl_xy[ck], cyl_r[ck], cyl_h[ck]), int(keep.sum())


def stream(scene, sensor, n, speed, seed=0, rate_hz=10.0, yaw_rate_deg=2.0):
    step = speed / rate_hz
    x0 = -0.5 * step * (n - 1)
    scans, poses = [], []
    for i in range(n):
        yaw = np.deg2rad(yaw_rate_deg) * np.sin(0.35 * i)
        pose = synth.pose_matrix([x0 + step * i, 0.4 * np.sin(0.2 * i), 0.0], [0.0, 0.0, yaw])    # workloads.make_odometry_stream
        scans.append(synth.scan(scene, sensor, pose, 5000 + 31 * seed + i))
        poses.append(pose)
    return scans, poses


def kitti_prefilter(cloud):
    p = O.default_prefilter_params()
    p.use_distance_filter, p.distance_near_thresh, p.distance_far_thresh = 1, 0.1, 100.0
    p.downsample_method, p.downsample_resolution = 1, 0.25
    p.outlier_removal_method, p.radius_radius, p.radius_min_neighbors = 2, 0.5, 2
    out = O.prefilter(cloud, p)
    return synth.to_xyzi(out[:, :3], out[:, 3])


def run(scans, poses, method="NDT_OMP", resolution=1.0, line_search=False, prefilter=False, eps=0.01, max_corr=2.5):
    p = O.default_params(O.HGS_NDT_OMP if method == "NDT_OMP" else O.HGS_FAST_GICP)
    p.resolution, p.transformation_epsilon, p.ndt_line_search = resolution, eps, 1 if line_search else 0
    if method != "NDT_OMP":
        p.max_correspondence_distance = max_corr
    od = ScanMatchingOdometry(O.OracleRegistration(p), keyframe_delta_trans=5.0, keyframe_delta_angle=2.0, keyframe_delta_time=10000.0,
                              downsample=kitti_prefilter if prefilter else None)
    t0 = time.perf_counter()
    est, its, conv, pts = [], [], 0, []
    for i, c in enumerate(scans):
        est.append(od.matching(0.1 * i, c))
        if od.last_result is not None and i > 0:
            its.append(od.last_result.iterations)
            conv += int(od.last_result.converged)
    gt0 = np.linalg.inv(poses[0])
    err = [synth.pose_error(e, gt0 @ q) for e, q in zip(est, poses)]
    travelled = float(np.linalg.norm((gt0 @ poses[-1])[:3, 3]))
    return dict(final=err[-1][0], rmse=float(np.sqrt(np.mean([e[0] ** 2 for e in err]))), rot=float(np.sqrt(np.mean([e[1] ** 2 for e in err]))), its=float(np.mean(its)),
                conv=conv, n=len(scans) - 1, travelled=travelled, keyframes=od.num_keyframes, s=time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=60)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_ndt_tracking.md"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    O.set_num_threads(os.cpu_count() or 1)
    scenes = {"corridor": (synth.make_scene(0), 80)}
    scenes["uniform"] = uniform_scene(0)
    rows = []
    speeds = (3.0,) if a.quick else (1.0, 2.0, 3.0, 5.0, 8.0)
    for speed in speeds:
        for scene_name, (scene, n_boxes) in scenes.items():
            scans, poses = stream(scene, "HDL-64E", a.scans, speed)
            npts = int(np.mean([len(s) for s in scans]))
            variants = [("NDT_OMP", dict()), ("NDT_OMP", dict(prefilter=True))]
            if speed in (3.0, 8.0):
                variants += [("NDT_OMP", dict(line_search=True)), ("NDT_OMP", dict(prefilter=True, line_search=True)), ("NDT_OMP", dict(resolution=2.0)),
                             ("FAST_GICP", dict(prefilter=True, eps=0.1, max_corr=2.0))]
            if speed == 3.0:
                variants += [("NDT_OMP", dict(resolution=1.5))]
            if speed == 5.0:
                variants += [("NDT_OMP", dict(resolution=2.0)), ("NDT_OMP", dict(resolution=2.0, prefilter=True))]
            for method, kw in variants:
                r = run(scans, poses, method=method, **kw)
                rows.append((speed, scene_name, n_boxes, npts, method, kw, r))
                print(f"{speed} m/s {scene_name:9s} {method:9s} {kw}: final {r['final']:.3f} m of {r['travelled']:.1f} m, rmse {r['rmse']:.3f}, {r['its']:.1f} iterations, "
                      f"{r['conv']}/{r['n']} converged, {r['s']:.0f} s", flush=True)
    with open(a.out, "w") as fh:
        fh.write("# NDT_OMP odometry on the synthetic config-3 stream: what makes it drift (round 6, CPU oracle, scripts/r06_ndt_tracking.py)\n\n")
        fh.write(f"HDL-64E, {a.scans} sweeps at 10 Hz, frame-to-keyframe with the KITTI launch file's keyframe rule (5 m / 2 rad), guess = previous result "
                 "(apps/scan_matching_odometry_nodelet.cpp:210).  `keeps track` = final error below 2 % of the distance travelled.\n\n")
        fh.write("**Reading.**  It is the basin of ndt_omp's Newton step (step length <= 0.1, no line search, guess = the previous result — no motion model) on noise-free planar "
                 "scenes, not the corridor, not a kernel and not the restatement's step rule: at 8 m/s (0.8 m per sweep) the estimate stays at the origin in EVERY variant — both "
                 "scenes, with and without the KITTI prefilter, with and without the opt-in line search, resolution 1 and 2 — because the cells of a wall scanned with 2 cm noise are a "
                 "few centimetres thin and a point 0.8 m off contributes nothing to the gradient; 3 m/s is the edge of the basin (which variants track is a coin toss); at <= 2 m/s "
                 "every variant tracks.  FAST_GICP with the KITTI launch file's parameters tracks at every speed.  The device reproduces the oracle stream bit for bit "
                 "(tests/test_odometry.py), so the table holds for the HIP path.  Consequence: bench.py --config 3 quotes `value` at 2 m/s and carries the 8 m/s stream as `at_8_mps`.\n\n")
        fh.write("| speed | scene | points / sweep | engine | prefilter | line search | resolution | final error [m] | travelled [m] | RMSE [m] | RMSE rot [rad] | iterations | converged | keyframes | keeps track |\n")
        fh.write("|---|---|---:|---|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---|\n")
        for speed, scene_name, n_boxes, npts, method, kw, r in rows:
            fh.write(f"| {speed:g} m/s | {scene_name} ({n_boxes} boxes) | {npts} | {method} | {'KITTI launch' if kw.get('prefilter') else 'none'} | {'on' if kw.get('line_search') else 'off (ndt_omp)'} | "
                     f"{kw.get('resolution', 1.0):g} | {r['final']:.3f} | {r['travelled']:.1f} | {r['rmse']:.3f} | {r['rot']:.4f} | {r['its']:.1f} | {r['conv']}/{r['n']} | {r['keyframes']} | "
                     f"{'yes' if r['final'] < 0.02 * r['travelled'] else 'NO'} |\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()
