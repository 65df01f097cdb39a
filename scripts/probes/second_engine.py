"""Is the SECOND engine of a process slower?  (bench.py's PLANE sub-record runs 9.1 ms per step as the third engine of the process, 8.3 ms alone.)
python scripts/probes/second_engine.py [first_method|none] — times a PLANE FAST_GICP batch after an optional first engine has run and been closed."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdl_graph_slam_amd import workloads, _lib as L
from hdl_graph_slam_amd.registrations import select_registration_method

if os.environ.get("PROBE_TORCH"):
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
first = sys.argv[1] if len(sys.argv) > 1 else "none"
keep = len(sys.argv) > 2 and sys.argv[2] == "keep"
wl = workloads.make_loop_closure_set("HDL-64E", 0, 64)

def run(pnh, n, label):
    reg = select_registration_method(pnh, device_id=0)
    tgt = reg.upload(wl.target)
    cands = [reg.upload(c) for c in wl.candidates]
    def step():
        tgt.invalidate()
        for c in cands:
            c.invalidate()
        reg.setInputTarget(tgt)
        return reg.loop_match_batch(cands, wl.guesses, L.DBL_MAX)
    for _ in range(5):
        step()
    reg.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step()
    reg.synchronize()
    print(f"{label}: {(time.perf_counter() - t) / n * 1e3:.3f} ms per step", flush=True)
    if os.environ.get("PROBE_PROFILE") and "first" in label:   # what bench.py does after a timed region: two steps with the stage timers (timed HIP events) on
        reg.profile_enable(True)
        reg.profile_read(reset=True)
        step(), step()
        reg.profile_read(reset=True)
        reg.profile_enable(False)
        t = time.perf_counter()
        for _ in range(n):
            step()
        reg.synchronize()
        print(f"{label}, after a profiled phase on the same engine: {(time.perf_counter() - t) / n * 1e3:.3f} ms per step", flush=True)
    return reg, tgt, cands

held = None
for f in first.split("+")[:-1]:   # "A+B": engines A, then B (the loop below), then the PLANE engine
    pnh0 = {"registration_method": "NDT_OMP", "reg_resolution": 1.0} if f == "NDT_OMP" else {"registration_method": "FAST_GICP"}
    reg0, tgt0, cands0 = run(pnh0, 10, f"engine before ({f})")
    for c in cands0:
        c.close()
    tgt0.close()
    reg0.close()
first = first.split("+")[-1]
if first != "none":
    pnh = {"registration_method": "NDT_OMP", "reg_resolution": 1.0} if first == "NDT_OMP" else {"registration_method": "FAST_GICP"}
    if first == "PLANE":
        pnh["reg_regularization_method"] = "PLANE"
    reg, tgt, cands = run(pnh, 10, f"first engine ({first})")
    if keep:
        held = (reg, tgt, cands)
    else:
        for c in cands:
            c.close()
        tgt.close()
        reg.close()
run({"registration_method": "FAST_GICP", "reg_regularization_method": "PLANE"}, 20, f"PLANE engine after first={first}{' (still open)' if keep else ''}")
