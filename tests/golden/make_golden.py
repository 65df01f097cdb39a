#!/usr/bin/env python3
"""Generates tests/golden/*.npz — known-answer vectors for the hot path.

The reference ships no tests or golden vectors, so these pins are created here: seeded synthetic clouds +
the CPU oracle's outputs (per-stage and per-iteration).  They freeze the oracle's behaviour (regressions in oracle/ show up
as a diff) and give the HIP path fixed inputs/outputs that travel to the GPU box.  Re-run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import oracle as O  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402


def main():
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.3)
    tx, sx = synth.xyz_of(tgt), synth.xyz_of(src)
    out = {"target_xyz": tx, "source_xyz": sx, "T_gt": T}
    # ---- FAST_GICP (factory defaults, registrations.cpp:27-36)
    p = O.default_params(O.HGS_FAST_GICP)
    o = O.OracleRegistration(p)
    o.setInputTarget(tx)
    o.setInputSource(sx)
    r = o.align(np.eye(4))
    out.update(gicp_final=r.matrix(), gicp_converged=r.converged, gicp_iterations=r.iterations, gicp_lm_tries=r.lm_tries, gicp_error=r.error,
               gicp_trace=o.trace())
    H, b, err, corr = o.gicp_linearize(np.eye(4))
    out.update(gicp_H_identity=H, gicp_b_identity=b, gicp_err_identity=err, gicp_corr_identity=corr)
    out["target_cov"] = O.covariances(tx, 20)
    q = sx[::5]
    idx, d2 = o.nn_target(q)
    out.update(nn_queries=q, nn_idx=idx, nn_d2=d2)
    ranges = np.array([np.finfo(np.float64).max, 4.0, 0.25])
    fit = []
    for mr in ranges:
        fit.append([o.getFitnessScore(mr, T=r.matrix()), o.last_num_inliers])
    out.update(fitness_ranges=ranges, fitness=np.array(fit))
    # ---- NDT_OMP DIRECT7, resolution 1.0 (launch files) from an odometry-like guess
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution = 1.0
    o = O.OracleRegistration(p)
    o.setInputTarget(tx)
    o.setInputSource(sx)
    ijk, mean, icov, npts = o.ndt_cells()
    out.update(ndt_cells_ijk=ijk, ndt_cells_mean=mean, ndt_cells_icov=icov, ndt_cells_n=npts)
    p0 = np.array([T[0, 3] + 0.03, T[1, 3] - 0.02, T[2, 3], 0.004, -0.006, np.arctan2(T[1, 0], T[0, 0]) + 0.003])
    s, g, Hn = o.ndt_derivatives(p0)
    out.update(ndt_p0=p0, ndt_score_p0=s, ndt_g_p0=g, ndt_H_p0=Hn)
    # a guess from which the oracle settles quickly (see tests/test_device_logic_host.py on NDT's chaotic iteration)
    best = None
    for off in ([0.02, 0.01, 0.0, 0.002], [0.1, -0.05, 0.0, 0.01], [0.05, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]):
        guess = (T @ synth.pose_matrix(off[:3], [0, 0, off[3]])).astype(np.float32)
        rn = o.align(guess)
        if best is None or rn.iterations < best[1].iterations:
            best = (guess, rn, o.trace())
    out.update(ndt_guess=best[0], ndt_final=best[1].matrix(), ndt_iterations=best[1].iterations, ndt_trace=best[2])
    np.savez_compressed(os.path.join(HERE, "vlp16_pair_seed1.npz"), **out)
    print("wrote", os.path.join(HERE, "vlp16_pair_seed1.npz"), {k: np.shape(v) for k, v in out.items()})
    print("gicp iterations", r.iterations, "ndt iterations", best[1].iterations)


if __name__ == "__main__":
    main()
