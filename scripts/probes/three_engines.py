"""Three engines alive in ONE process (the nodelet manager hosts the odometry engine and the loop-closure engine; bench.py creates several): does the
third engine's batch run as fast as the first engine's did when it was alone?
    GPU_MAX_HW_QUEUES=16 python scripts/probes/three_engines.py [n_candidates] [n_distinct]
prints "alone <ms> third <ms> lanes_budget <queues>".  The library counts the streams it creates against the process's hardware-queue budget
(GPU_MAX_HW_QUEUES, else HIP's 4) and opens only as many lanes as there is room for: with room (16) the third engine has its four lanes on queues of
their own; without (4) it runs on one lane (a few per cent) instead of sharing queues (8 %, round 4)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdl_graph_slam_amd import workloads, _lib as L  # noqa: E402
from hdl_graph_slam_amd.registrations import select_registration_method  # noqa: E402

n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = workloads.make_loop_closure_set("HDL-64E", 0, n_cand, n_distinct=n_dist)


def engine(pnh):
    reg = select_registration_method(pnh, device_id=0)
    tgt = reg.upload(wl.target)
    cands = [reg.upload(c) for c in wl.candidates]

    def step():
        tgt.invalidate()
        for c in cands:
            c.invalidate()
        reg.setInputTarget(tgt)
        return reg.loop_match_batch(cands, wl.guesses, L.DBL_MAX)
    return reg, step


def rate(reg, step, n=12):
    for _ in range(4):
        step()
    ts = []
    for _ in range(n):
        reg.synchronize()
        t = time.perf_counter()
        step()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


a, step_a = engine({"registration_method": "FAST_GICP"})
alone = rate(a, step_a)
b, step_b = engine({"registration_method": "NDT_OMP", "reg_resolution": 1.0})
step_b()                      # (its lanes exist now)
c, step_c = engine({"registration_method": "FAST_GICP"})
third = rate(c, step_c)
again = rate(a, step_a)       # the first engine, with the other two alive
print(f"alone {alone:.3f} third {third:.3f} first_again {again:.3f} lanes_budget {os.environ.get('GPU_MAX_HW_QUEUES', '4 (HIP default)')}")
