#!/usr/bin/env python3
"""Round 5, the one structural experiment on the exact tree walk: would a WIDER node — an 8- or 16-ary record, i.e. 3 or 4
binary levels decided per dependent fetch instead of 2 — shorten the seeded 1-NN packet walk of k_gicp_linearize?

Host simulation (tests/emul/emul.cpp: sim_wave_walk_wide, 64 lanes in lock-step, exactness checked against the per-lane search) on the
benchmark's own clouds (HDL-64E, SURVEY 8d candidate set, scene seed 0).  Per 64-query packet:
   steps      = nodes entered = dependent record fetches (the quantity the 36 % SQ_WAIT_INST_ANY of the kernel is made of)
   box tests  = boxes the wave evaluates (records x arity + one re-test per popped child) — the VALU work of the walk
   leaves     = 8-point leaves visited
for three situations: the unseeded first linearisation (at the guess), a seeded linearisation after a LARGE pose move (seeds from the guess, query at
the ground truth: iteration 2) and after a SMALL move (1 cm / 0.05 deg: the late iterations).
Build criterion of the verdict: >= 25 % fewer dependent steps at <= 10 % more box tests.

    python scripts/walk_wide_nodes.py [n_candidates]          -> profiles/r05_ab_wide_nodes.log (stdout)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
from emul import emul  # noqa: E402
from hdl_graph_slam_amd import synth, workloads  # noqa: E402


def main():
    n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    wl = workloads.make_loop_closure_set("HDL-64E", scene_seed=0, n_candidates=n_cand, n_distinct=n_cand)
    p = O.default_params(O.HGS_FAST_GICP)
    bound2 = np.float32(2.5 ** 2)
    L = emul.lib()
    L.emul_sim_box_tests.restype = __import__("ctypes").c_longlong
    names = {1: "4-ary, direct pops + in-place leaves (the kernel's walk; quad fetch not modelled)", 2: "wide S=2 (4-ary on the generic stack walk: the control)",
             3: "wide S=3 (8-ary record, 192 B)", 4: "wide S=4 (16-ary record, 384 B)"}
    tot = {v: {} for v in names}
    for c in range(n_cand):
        e = emul.EmulRegistration(p)
        e.setInputTarget(wl.target)
        e.setInputSource(wl.candidates[c])
        guess, gt = np.asarray(wl.guesses[c], np.float64), np.asarray(wl.T_gt[c], np.float64)
        small = gt @ synth.pose_matrix([0.01, 0.004, 0.0], [0.0, 0.0, 0.0009])
        cases = (("unseeded @guess", guess, None), ("seeded, large move (guess -> ground truth)", gt, guess), ("seeded, small move (1 cm)", small, gt))
        for case, T, seed_pose in cases:
            for v in names:
                if seed_pose is not None:
                    L.emul_set_sim_variant(0)
                    e.gicp_linearize(seed_pose)       # leaves the correspondences at seed_pose behind as seeds
                L.emul_set_sim_variant(v)
                L.emul_sim_box_tests(1)
                groups, leaves, waves, mism = e.walk_stats(T, bound2, seed_pose is not None)
                boxes = L.emul_sim_box_tests(1) if v >= 2 else 4 * groups
                assert mism == 0, (case, v, mism)
                a = tot[v].setdefault(case, np.zeros(4))
                a += (groups, boxes, leaves, waves)
        L.emul_set_sim_variant(0)
    print(f"HDL-64E, {n_cand} candidate(s) x ~{len(wl.candidates[0])} points against a ~{len(wl.target)}-point target, max correspondence distance 2.5 m; per 64-query packet")
    for case in next(iter(tot.values())):
        print(f"\n{case}")
        base = tot[1][case]
        print(f"  {'variant':78s} {'steps':>8s} {'box tests':>10s} {'leaves':>8s}   steps vs kernel   box tests vs kernel")
        for v, name in names.items():
            g, b, lv, w = tot[v][case]
            print(f"  {name:78s} {g / w:8.1f} {b / w:10.1f} {lv / w:8.1f}   {g / base[0] - 1:+14.0%}   {b / base[1] - 1:+18.0%}")
    print("\n(the kernel's quad fetch already turns ~53 seeded steps into ~29 memory waits: the four children of a node arrive with one 512-byte load and are parked in LDS, "
          "DESIGN.md section 4; a 16-ary record would need 6 KB for the same trick)")


if __name__ == "__main__":
    main()
