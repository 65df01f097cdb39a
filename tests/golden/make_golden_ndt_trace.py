#!/usr/bin/env python3
"""Generates tests/golden/vlp16_ndt_long_run.npz: a LONG NDT run (ndt_omp's clamped Newton iteration from a guess 0.3 m /
0.02 rad off, tens of iterations) as a per-iteration trace of the oracle in its exact-sum mode — parameters, score and step
length of every iteration, iteration / pass counts and the final float 4x4.  The device backend implements the same sum, so
it must end on the very same float matrix (tests/test_golden.py).  Re-run:  python tests/golden/make_golden_ndt_trace.py"""
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
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution = 1.0
    out = {"target_xyz": tx, "source_xyz": sx}
    best = None
    for off in ([0.3, 0.1, 0.0, 0.02], [0.02, 0.01, 0.0, 0.002], [-0.2, 0.15, 0.05, -0.015]):
        guess = (T @ synth.pose_matrix(off[:3], [0, 0, off[3]])).astype(np.float32)
        o = O.OracleRegistration(p).set_ndt_sum_mode(1)
        o.setInputTarget(tx)
        o.setInputSource(sx)
        r = o.align(guess)
        if best is None or r.iterations > best[1].iterations:
            serial = O.OracleRegistration(p)
            serial.setInputTarget(tx)
            serial.setInputSource(sx)
            best = (guess, r, o.trace(), serial.align(guess))
    guess, r, trace, rs = best
    out.update(guess=guess, trace=trace, iterations=r.iterations, passes=r.lm_tries, converged=r.converged,
               final=np.array(r.final_transformation, np.float32), final_serial_sum=np.array(rs.final_transformation, np.float32))
    np.savez_compressed(os.path.join(HERE, "vlp16_ndt_long_run.npz"), **out)
    print("wrote vlp16_ndt_long_run.npz: iterations", r.iterations, "passes", r.lm_tries, "trace", trace.shape)


if __name__ == "__main__":
    main()
