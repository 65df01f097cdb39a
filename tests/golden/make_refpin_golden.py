#!/usr/bin/env python3
"""Generates tests/golden/refpin_v1.npz and tests/golden/refpin_keyframes/ — vectors produced by REFERENCE CODE RUN HERE: the unmodified
src/hdl_graph_slam/information_matrix_calculator.cpp and src/hdl_graph_slam/keyframe.cpp of /root/reference, compiled against tests/mock_*
(tests/refpin_build.py, tests/cpp/refpin_main.cpp).  Needs /root/reference (the build container); the GPU box only reads the committed vectors.

    fit_*      calc_fitness_score (:49-80) on committed clouds (tests/golden/vlp16_pair_seed1.npz, vlp16_next_rows_v3.npz) at several poses and
               max_range values incl. the declaration's default (DBL_MAX), the squared-vs-unsquared comparison (1.0 vs 0.25), a range with no
               inlier (-> DBL_MAX) and an empty source; once more with PCL 1.8's transform order (`fit_score_pcl18`);
    big_*      the same on one seeded HDL-32E pair of the size the loop-closure batch uses (inputs by sha256: synth is deterministic);
    inf_*      calc_information_matrix (:25-47) for several rosparam sets;
    keyframes  directories written by KeyFrame::save (:21-58) + what KeyFrame::load (:60-145) reads back from them and from directories written by
               hdl_graph_slam_amd.keyframe_io (kf_*_loaded_by_reference).
Re-run: python tests/golden/make_refpin_golden.py"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import refpin_build as RB  # noqa: E402
from hdl_graph_slam_amd import synth, keyframe_io as K  # noqa: E402

DBL_MAX = float(np.finfo(np.float64).max)
RANGES = [DBL_MAX, 4.0, 1.0, 0.25, 1e-9]          # the first is passed as "max" = the declaration's default argument
KF_DIR = os.path.join(HERE, "refpin_keyframes")


def small_clouds():
    z = np.load(os.path.join(HERE, "vlp16_pair_seed1.npz"))
    n = np.load(os.path.join(HERE, "vlp16_next_rows_v3.npz"))
    def xyzi(a):
        return synth.to_xyzi(a[:, :3], a[:, 3])
    return {"pair_target": synth.to_xyzi(z["target_xyz"]), "pair_source": synth.to_xyzi(z["source_xyz"]), "kf0": xyzi(n["kf0_xyzi"]), "kf1": xyzi(n["kf1_xyzi"]),
            "kf2": xyzi(n["kf2_xyzi"]), "empty": synth.to_xyzi(np.zeros((0, 3), np.float32))}, z, n


def fitness_cases():
    clouds, z, n = small_clouds()
    poses = n["poses"].astype(np.float64)
    rel = lambda a, b: np.linalg.inv(poses[a]) @ poses[b]
    cases = []
    for name, T in (("gt", z["T_gt"]), ("gicp_final", z["gicp_final"].astype(np.float64)), ("identity", np.eye(4)),
                    ("perturbed", z["T_gt"] @ synth.pose_matrix([0.3, -0.2, 0.05], [0.01, -0.02, 0.05]))):
        for r in RANGES:
            cases.append(("pair_target", "pair_source", name, T, r))
    for a, b in ((0, 1), (1, 2), (2, 0)):
        for r in (DBL_MAX, 4.0):
            cases.append((f"kf{a}", f"kf{b}", f"rel{a}{b}", rel(a, b), r))
    cases.append(("pair_target", "empty", "identity", np.eye(4), DBL_MAX))
    return clouds, cases


def big_pair():
    tgt, src, T = synth.make_pair("HDL-32E", 4)
    return tgt, src, T


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pose_args(T):
    return [repr(float(v)) for v in np.asarray(T, np.float64).reshape(16)]


def run_fitness(tmp, c1, c2, T, r, variant=""):
    out = RB.run(["fitness", os.path.join(tmp, c1 + ".bin"), os.path.join(tmp, c2 + ".bin"), "max" if r == DBL_MAX else repr(float(r)), *pose_args(T)], variant)
    return float(out.split()[1])


INF_PARAMS = [{}, {"use_const_inf_matrix": "true"}, {"var_gain_a": "5.0", "fitness_score_thresh": "2.5"},
              {"min_stddev_x": "0.05", "max_stddev_x": "2.0", "min_stddev_q": "0.01", "max_stddev_q": "0.5", "const_stddev_x": "0.3"}]


def keyframe_specs():
    clouds, z, n = small_clouds()
    def hexm(M):
        return [float(v).hex() for v in np.asarray(M, np.float64).reshape(-1)]
    full = dict(stamp=[1600000123, 456789012], estimate=synth.pose_matrix([1.5, -20.25, 0.3], [0.01, -0.02, 1.2]), odom=synth.pose_matrix([100.123456789, -2.5, 0.0], [0.0, 0.0, -0.7]),
                accum_distance=123.456789012, floor_coeffs=[0.001, -0.02, 0.9998, 1.85], utm_coord=[345678.125, 3987654.5, 12.75], acceleration=[0.1, -9.81, 0.25],
                orientation=[0.9238795, 0.0, 0.3826834, 0.0], id=42, cloud="kf0")
    minimal = dict(stamp=[0, 7], estimate=np.eye(4), odom=np.eye(4), accum_distance=0.0, id=0, cloud="kf1")
    odd = dict(stamp=[4294967295, 999999999], estimate=synth.pose_matrix([-1e-7, 1234567.891, -0.000123456789], [3.1, -1.5, 0.0004]),
               odom=synth.pose_matrix([1e10, -1e-10, 5.0], [-0.3, 0.2, 2.9]), accum_distance=98765.4321, utm_coord=[-0.0, 1e-300, 1e300], id=123456, cloud="empty")
    out = []
    for name, s in (("000000", full), ("000001", minimal), ("000002", odd)):
        s = dict(s)
        s["estimate"], s["odom"] = np.asarray(s["estimate"], np.float64), np.asarray(s["odom"], np.float64)
        out.append((name, s))
    return clouds, out


def spec_text(s, cloud_path):
    def f(v):
        return " ".join(float(x).hex() for x in np.asarray(v, np.float64).reshape(-1))
    lines = [f"stamp {s['stamp'][0]} {s['stamp'][1]}", "estimate " + f(s["estimate"]), "odom " + f(s["odom"]), "accum_distance " + f([s["accum_distance"]])]
    for k in ("floor_coeffs", "utm_coord", "acceleration", "orientation"):
        if k in s:
            lines.append(k + " " + f(s[k]))
    lines += [f"id {s['id']}", f"cloud {cloud_path}"]
    return "\n".join(lines) + "\n"


def record_of(s, cloud):
    return K.KeyFrameRecord(tuple(s["stamp"]), s["estimate"], s["odom"], s["accum_distance"], cloud, s["id"],
                            *[np.asarray(s[k], np.float64) if k in s else None for k in ("floor_coeffs", "utm_coord", "acceleration", "orientation")])


def generate(out_npz, out_kf_dir):
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        clouds, cases = fitness_cases()
        for k, c in clouds.items():
            c.tofile(os.path.join(tmp, k + ".bin"))
        res["fit_cloud1"] = np.array([c[0] for c in cases])
        res["fit_cloud2"] = np.array([c[1] for c in cases])
        res["fit_pose_name"] = np.array([c[2] for c in cases])
        res["fit_pose"] = np.stack([np.asarray(c[3], np.float64) for c in cases])
        res["fit_max_range"] = np.array([c[4] for c in cases])
        res["fit_score"] = np.array([run_fitness(tmp, c[0], c[1], c[3], c[4]) for c in cases])
        res["fit_score_pcl18"] = np.array([run_fitness(tmp, c[0], c[1], c[3], c[4], variant="pcl18") for c in cases])
        # the loop-closure-sized pair
        tgt, src, T = big_pair()
        tgt.tofile(os.path.join(tmp, "big_t.bin"))
        src.tofile(os.path.join(tmp, "big_s.bin"))
        res["big_sha256"] = np.array([sha(tgt), sha(src)])
        res["big_n"] = np.array([len(tgt), len(src)])
        res["big_pose"] = np.stack([T, T @ synth.pose_matrix([0.4, 0.1, 0.0], [0.0, 0.0, 0.03])])
        res["big_max_range"] = np.array([DBL_MAX, 4.0])
        res["big_score"] = np.array([[run_fitness(tmp, "big_t", "big_s", P, r) for r in res["big_max_range"]] for P in res["big_pose"]])
        # information matrices
        inf_rows, inf_fit = [], []
        T = res["fit_pose"][0]
        for prm in INF_PARAMS:
            for pose_i in (0, 15):   # the ground-truth pose and the perturbed one of the VLP-16 pair, both at the default range
                P = res["fit_pose"][pose_i]
                out = RB.run(["infomat", os.path.join(tmp, "pair_target.bin"), os.path.join(tmp, "pair_source.bin"), *pose_args(P), *[f"{k}={v}" for k, v in prm.items()]])
                tok = out.split()
                assert tok[:3] == ["infomat", "6", "6"]
                inf_rows.append(np.array([float(v) for v in tok[3:]]).reshape(6, 6))
                inf_fit.append(res["fit_score"][pose_i])      # (cases 0 and 15 are DBL_MAX-range cases)
        res["inf_params"] = np.array([json.dumps(p) for p in INF_PARAMS for _ in (0, 1)])
        res["inf_pose_case"] = np.array([i for _ in INF_PARAMS for i in (0, 15)])
        res["inf_fitness"] = np.array(inf_fit)
        res["inf_matrix"] = np.stack(inf_rows)
        # keyframe directories
        if os.path.isdir(out_kf_dir):
            shutil.rmtree(out_kf_dir)
        os.makedirs(out_kf_dir)
        kclouds, specs = keyframe_specs()
        loaded_own, loaded_ours, spec_json = [], [], {}
        for name, s in specs:
            spec_path = os.path.join(tmp, name + ".spec")
            with open(spec_path, "w") as fh:
                fh.write(spec_text(s, os.path.join(tmp, s["cloud"] + ".bin")))
            RB.run(["kf_save", os.path.join(out_kf_dir, name), spec_path])
            loaded_own.append(RB.run(["kf_load", os.path.join(out_kf_dir, name), s["id"], os.path.join(tmp, "own.bin")]))
            assert np.fromfile(os.path.join(tmp, "own.bin"), synth.POINT_XYZI_DTYPE).tobytes() == _canonical(kclouds[s["cloud"]]).tobytes()
            ours = os.path.join(tmp, "ours_" + name)
            K.save_keyframe(ours, record_of(s, kclouds[s["cloud"]]))
            loaded_ours.append(RB.run(["kf_load", ours, s["id"], os.path.join(tmp, "ours.bin")]))
            spec_json[name] = {k: (v if isinstance(v, (int, str)) else [float(x).hex() for x in np.asarray(v, np.float64).reshape(-1)]) for k, v in s.items()}
        res["kf_names"] = np.array([n for n, _ in specs])
        res["kf_spec_json"] = np.array(json.dumps(spec_json))
        res["kf_loaded_by_reference"] = np.array(loaded_own)
        res["kf_ours_loaded_by_reference"] = np.array(loaded_ours)
    np.savez_compressed(out_npz, **res)
    return res


def _canonical(cloud):
    """What a PointXYZI record looks like after a trip through cloud.pcd: x, y, z, intensity kept, data[3] = 1, padding zero."""
    c = np.zeros(len(cloud), synth.POINT_XYZI_DTYPE)
    for f in ("x", "y", "z", "intensity"):
        c[f] = cloud[f]
    c["w"] = 1.0
    return c


if __name__ == "__main__":
    r = generate(os.path.join(HERE, "refpin_v1.npz"), KF_DIR)
    print("fitness", r["fit_score"])
    print("pcl18 rel diff", np.max(np.abs(r["fit_score_pcl18"] / r["fit_score"] - 1)))
    print("big", r["big_score"], r["big_n"])
    print("wrote", os.path.getsize(os.path.join(HERE, "refpin_v1.npz")), "bytes")
