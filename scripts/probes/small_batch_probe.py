"""Loop-closure batches of the size a KITTI run produces — B candidate keyframes of ~11-14 k points (prefiltered sweeps) against one keyframe — with the LM rounds in
four launches per lane (batches above 4 problems, as shipped until now) and in two (engine options fused_rounds_max_problems / fused_rounds_max_blocks).  Prints
milliseconds per detection (p50) and whether the records are identical bits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdl_graph_slam_amd import synth  # noqa: E402
from hdl_graph_slam_amd.registrations import select_registration_method  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tgt, src, T = synth.make_pair("HDL-64E", 3, downsample=0.25)
rng = np.random.default_rng(5)
for reg_method in ("FROBENIUS", "PLANE"):
    reg = select_registration_method({"registration_method": "FAST_GICP", "reg_regularization_method": reg_method}, device_id=0)
    reg.setInputTarget(tgt)
    for B in (6, 12, 24, 48, 64):
        clouds = [reg.upload(src[rng.permutation(len(src))[: len(src) - 37 * k]]) for k in range(B)]
        guesses = [np.asarray(synth.pose_matrix(rng.normal(0, 0.15, 3), rng.normal(0, 0.01, 3)), np.float32) for _ in range(B)]
        ref = None
        row = []
        for blocks in (0, 100000, 0, 100000):
            reg.set_option("fused_rounds_max_blocks", blocks)
            ts = []
            for rep in range(reps):
                for c in clouds:
                    c.invalidate()  # cold: index + covariances rebuilt, as the reference does per setInputSource
                t0 = time.perf_counter()
                rec, best = reg.loop_match_batch(clouds, guesses)
                ts.append(time.perf_counter() - t0)
            bits = (rec.tobytes(), best)
            if ref is None:
                ref = bits
            row.append(f"{'two' if blocks else 'four'}: {1e3 * float(np.median(ts[3:])):.3f} ms{'' if bits == ref else ' MISMATCH'}")
        print(reg_method, "B", B, "points", len(src), "mean iterations", float(np.mean(rec["iterations"])), "|", " | ".join(row))
        for c in clouds:
            c.close()
    reg.close()
