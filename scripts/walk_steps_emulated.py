#!/usr/bin/env python3
"""Packet-walk steps of the PRODUCT kernels, counted by the host emulation (tests/emul): one step = one 128-byte record
(group of boxes or leaf) fetched by a wave.  Prints steps per wave for the covariance pass, every linearisation of one GICP
registration (first unseeded, then seeded by the previous correspondences) and the fitness pass — the figures DESIGN.md
section 4 / 9 reason with, taken from the code that runs on the GPU instead of a separate simulator.

  python scripts/walk_steps_emulated.py [--sensor HDL-32E] [--downsample 0.25]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from emul import simt  # noqa: E402
from hdl_graph_slam_amd import _lib as L, synth  # noqa: E402
from hdl_graph_slam_amd.registration import RegistrationHIP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sensor", default="HDL-32E")
    ap.add_argument("--downsample", type=float, default=0.25)
    ap.add_argument("--seed", type=int, default=4)
    args = ap.parse_args()
    L.LIB_PATH, L._lib = simt.build(), None
    lib = C.CDLL(L.LIB_PATH)
    buf = (C.c_ulonglong * 2)()

    def counters():
        lib.simt_read_counters(buf)
        return int(buf[0]), int(buf[1])

    tgt, src, T = synth.make_pair(args.sensor, args.seed, downsample=args.downsample or None)
    p = L.default_params(L.HGS_FAST_GICP)
    p.max_iterations = 0                       # one LM iteration per align call below (0 -> executes exactly one)
    reg = RegistrationHIP(p)
    counters()
    reg.setInputTarget(tgt), reg.setInputSource(src)
    reg.target_covariances(len(tgt))
    steps, waves = counters()
    print(f"points: target {len(tgt)}, source {len(src)}")
    print(f"index + covariance of the target: {steps} steps (3 walks per wave: k-th distance, gather, + tie walks), {steps / max(1, (len(tgt) + 63) // 64):.1f} per 64 queries")
    guess = np.eye(4)
    for it in range(5):
        r = reg.align(guess)
        steps, waves = counters()
        n_packets = (len(src) + 63) // 64
        label = "unseeded (incl. source covariances)" if it == 0 else "seeded"
        print(f"align #{it} (1 linearisation + its error pass{'' if it else ' + source index/covariance'}): {steps} steps, {steps / n_packets:.1f} per 64 source points  [{label}]")
        guess = r.matrix()
    reg.getFitnessScore()
    steps, waves = counters()
    print(f"fitness pass: {steps} steps, {steps / ((len(src) + 63) // 64):.1f} per 64 source points")
    reg.close()


if __name__ == "__main__":
    main()
