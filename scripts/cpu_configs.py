#!/usr/bin/env python3
"""CPU side of SURVEY §8(d): the oracle (OpenMP restatement of ndt_omp / fast_gicp) on configs 1 and 2 over the ten scene
seeds, at 1 and N threads: wall time of set_target / set_source+align / fitness (median of the seeds after one warm-up) and
pose error against the simulator's ground truth (mean +- std).  No GPU involved; the HIP path reproduces these poses to
<= 1e-5 m (tests), so the accuracy columns are the backend's too.

  python scripts/cpu_configs.py [--seeds 10] [--threads 1,8]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402

CONFIGS = {
    "cfg1 VLP-16 0.1 m voxel, NDT_OMP DIRECT7 res 1.0": ("VLP-16", 0.1, O.HGS_NDT_OMP, {"resolution": 1.0}),
    "cfg1 VLP-16 0.1 m voxel, NDT_OMP DIRECT7 res 0.5": ("VLP-16", 0.1, O.HGS_NDT_OMP, {"resolution": 0.5}),
    "cfg2 HDL-32E raw, FAST_GICP k 20 max_corr 2.5": ("HDL-32E", None, O.HGS_FAST_GICP, {}),
    "cfg2 HDL-32E raw, FAST_GICP k 20 max_corr 2.0": ("HDL-32E", None, O.HGS_FAST_GICP, {"max_correspondence_distance": 2.0}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--threads", default="1,%d" % (os.cpu_count() or 1))
    args = ap.parse_args()
    threads = [int(t) for t in args.threads.split(",")]
    pairs = {}
    out = []
    for name, (sensor, ds, method, over) in CONFIGS.items():
        for nt in threads:
            O.set_num_threads(nt)
            t_tgt, t_align, t_fit, et, er, its, conv, npts = [], [], [], [], [], [], [], []
            for seed in range(args.seeds):
                key = (sensor, ds, seed)
                if key not in pairs:
                    pairs[key] = synth.make_pair(sensor, seed, downsample=ds)
                tgt, src, T = pairs[key]
                p = O.default_params(method)
                for k, v in over.items():
                    setattr(p, k, v)
                o = O.OracleRegistration(p)
                if seed == 0:  # warm-up (thread pool, first touch)
                    o.setInputTarget(tgt), o.setInputSource(src), o.align(np.eye(4))
                    o = O.OracleRegistration(p)
                t0 = time.perf_counter()
                o.setInputTarget(tgt)
                t1 = time.perf_counter()
                o.setInputSource(src)
                r = o.align(np.eye(4))       # scan_matching_odometry_nodelet.cpp:210: the guess is the previous result; first frame = identity
                t2 = time.perf_counter()
                o.getFitnessScore()
                t3 = time.perf_counter()
                dt, dr = synth.pose_error(r.matrix(), T)
                t_tgt.append(t1 - t0), t_align.append(t2 - t1), t_fit.append(t3 - t2)
                et.append(dt), er.append(dr), its.append(r.iterations), conv.append(bool(r.converged)), npts.append(len(src))
            rec = {"config": name, "threads": nt, "seeds": args.seeds, "points": int(np.mean(npts)),
                   "set_target_ms": round(1e3 * float(np.median(t_tgt)), 2), "set_source_align_ms": round(1e3 * float(np.median(t_align)), 2),
                   "fitness_ms": round(1e3 * float(np.median(t_fit)), 2), "iterations_mean": round(float(np.mean(its)), 1), "converged": int(np.sum(conv)),
                   "translation_error_m": [round(float(np.mean(et)), 4), round(float(np.std(et)), 4)],
                   "rotation_error_rad": [round(float(np.mean(er)), 5), round(float(np.std(er)), 5)]}
            print(json.dumps(rec), flush=True)
            out.append(rec)
    return out


if __name__ == "__main__":
    main()
