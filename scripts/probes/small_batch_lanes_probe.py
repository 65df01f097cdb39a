"""Loop-closure batches of B candidate keyframes of ~11 k points (the shape a KITTI run produces) under 1 / 2 / 4 batch lanes, FAST_GICP and NDT_OMP:
milliseconds per detection (p50), records compared bit for bit across lane counts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdl_graph_slam_amd import synth  # noqa: E402
from hdl_graph_slam_amd.registrations import select_registration_method  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sensor, seed, ds = os.environ.get("SHAPE_SENSOR", "HDL-64E"), int(os.environ.get("SHAPE_SEED", "3")), float(os.environ.get("SHAPE_DS", "0.25"))
tgt, src, T = synth.make_pair(sensor, seed, downsample=ds if ds > 0 else None)
print(sensor, "points", len(src), "tiles per problem", (len(src) + 255) // 256)
rng = np.random.default_rng(5)
for method in os.environ.get("SHAPE_METHODS", "FAST_GICP,NDT_OMP").split(","):
    reg = select_registration_method({"registration_method": method, "reg_resolution": 1.0}, device_id=0)
    reg.setInputTarget(tgt)
    for B in [int(x) for x in os.environ.get("SHAPE_B", "6,12,24,48").split(",")]:
        clouds = [reg.upload(src[rng.permutation(len(src))[: len(src) - 37 * k]]) for k in range(B)]
        guesses = [np.asarray(T @ synth.pose_matrix(rng.normal(0, 0.05, 3), rng.normal(0, 0.003, 3)), np.float32) for _ in range(B)]
        ref = None
        row = []
        for lanes in (0, 1, 2, 4, 0):
            reg.set_option("batch_lanes", lanes)
            ts = []
            for rep in range(reps):
                for c in clouds:
                    c.invalidate()
                t0 = time.perf_counter()
                rec, best = reg.loop_match_batch(clouds, guesses)
                ts.append(time.perf_counter() - t0)
            bits = (rec.tobytes(), best)
            if ref is None:
                ref = bits
            row.append(f"{lanes}: {1e3 * float(np.median(ts[3:])):.3f}{'' if bits == ref else ' MISMATCH'}")
        print(method, "B", B, "mean iterations", round(float(np.mean(rec["iterations"])), 2), "| lanes: ms |", " | ".join(row))
        for c in clouds:
            c.close()
    reg.close()
