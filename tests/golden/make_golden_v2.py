#!/usr/bin/env python3
"""Generates tests/golden/vlp16_vgicp_prefilter_seed2.npz — known-answer vectors for the FAST_VGICP engine and the
prefilter (the rows added after vlp16_pair_seed1.npz).  Same rationale as make_golden.py: the reference has no golden
vectors, these freeze the oracle and give the HIP path fixed inputs/outputs.  Re-run: python tests/golden/make_golden_v2.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402


def main():
    out = {}
    # ---- prefilter: a raw VLP-16 sweep (every 3rd return) with intensities, a few far outliers and non-finite records
    scene = synth.make_scene(2)
    raw = synth.scan(scene, "VLP-16", synth.pose_matrix([0, 0, 0], [0, 0, 0]), 902)[::3]
    rng = np.random.default_rng(2)
    raw["intensity"] = rng.uniform(0, 255, len(raw)).astype(np.float32)
    extra = synth.to_xyzi((rng.uniform(-80, 80, (25, 3)) * [1, 1, 0.2]).astype(np.float32), rng.uniform(0, 255, 25))
    bad = synth.to_xyzi(np.array([[np.nan, 0, 0], [1, np.inf, 2]], np.float32))
    raw = np.concatenate([raw, extra, bad])
    out["raw_xyzi"] = np.stack([raw["x"], raw["y"], raw["z"], raw["intensity"]], axis=1)
    p = O.default_prefilter_params()                      # distance 1..100 m, VOXELGRID 0.1, STATISTICAL 20 / 1.0
    out["prefilter_default"] = O.prefilter(raw, p)
    p.downsample_resolution, p.outlier_removal_method, p.radius_radius, p.radius_min_neighbors = 0.25, 2, 0.5, 2
    out["prefilter_kitti_radius"] = O.prefilter(raw, p)  # KITTI launch resolution, RADIUS 0.5 / 2
    # ---- FAST_VGICP (registrations.cpp:48-56) on a small pair
    tgt, src, T = synth.make_pair("VLP-16", 2, downsample=0.3)
    tx, sx = synth.xyz_of(tgt), synth.xyz_of(src)
    out.update(target_xyz=tx, source_xyz=sx, T_gt=T)
    pv = O.default_params(O.HGS_FAST_VGICP)
    o = O.OracleRegistration(pv)
    o.setInputTarget(tx)
    o.setInputSource(sx)
    H, b, err, hits = o.gicp_linearize(np.eye(4))
    out.update(vgicp_H_identity=H, vgicp_b_identity=b, vgicp_err_identity=err, vgicp_hits_identity=hits)
    r = o.align(np.eye(4))
    out.update(vgicp_final=r.matrix(), vgicp_converged=r.converged, vgicp_iterations=r.iterations, vgicp_lm_tries=r.lm_tries, vgicp_error=r.error)
    path = os.path.join(HERE, "vlp16_vgicp_prefilter_seed2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
