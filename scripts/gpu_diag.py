#!/usr/bin/env python3
"""Diagnostics printed during a GPU-box visit: per-pass NDT parity level, per-stage timings at BASELINE sizes, CPU oracle
thread scaling.  Output goes to stdout (captured into gpurun_out/diag.log by scripts/gpu_round.sh)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O  # noqa: E402
from hdl_graph_slam_amd import synth, _lib as L  # noqa: E402
from hdl_graph_slam_amd.registration import RegistrationHIP  # noqa: E402


def hip(params):
    p = L.HgsParams()
    for name, _ in L.HgsParams._fields_:
        setattr(p, name, getattr(params, name))
    return RegistrationHIP(p)


def ndt_parity():
    for kind, (tgt, src, T) in {"vlp16": synth.make_pair("VLP-16", 2, downsample=0.1), "hdl32": synth.make_pair("HDL-32E", 4, downsample=0.25)}.items():
        p = O.default_params(O.HGS_NDT_OMP)
        p.resolution = 1.0
        e, o = hip(p), O.OracleRegistration(p)
        for r in (e, o):
            r.setInputTarget(tgt)
            r.setInputSource(src)
        for p6 in ([T[0, 3], T[1, 3], T[2, 3], 0.003, -0.004, 0.02], [T[0, 3] + 0.1, T[1, 3], T[2, 3], 3.14, 3.13, 3.1]):
            se, ge, He = e.ndt_derivatives(np.array(p6))
            so, go, Ho = o.ndt_derivatives(np.array(p6))
            print(f"ndt {kind}: score rel {abs(se - so) / abs(so):.2e} g rel {np.abs(ge - go).max() / np.abs(go).max():.2e} H rel {np.abs(He - Ho).max() / np.abs(Ho).max():.2e}")
        for off in ([0.02, 0.01, 0.0, 0.002], [0.05, 0.02, 0.0, 0.004], [0.1, -0.05, 0.0, 0.01]):
            g = T @ synth.pose_matrix(off[:3], [0, 0, off[3]])
            re, ro = e.align(g), o.align(g)
            dt, dr = synth.pose_error(re.matrix(), ro.matrix())
            print(f"ndt {kind} off {off}: it hip {re.iterations} oracle {ro.iterations}  diff {dt:.2e} m {dr:.2e} rad")
        e.close()


def stage_times():
    cases = {"hdl32 60k": synth.make_pair("HDL-32E", 5), "hdl64 120k": synth.make_pair("HDL-64E", 0), "dense 1M": synth.make_dense_pair(0, 1_000_000)}
    for name, (tgt, src, T) in cases.items():
        for method, pnh in ((L.HGS_FAST_GICP, {}), (L.HGS_NDT_OMP, {"resolution": 1.0})):
            p = L.default_params(method)
            if method == L.HGS_FAST_GICP and "dense" in name:
                p.max_correspondence_distance = 1.0
            for k, v in pnh.items():
                setattr(p, k, v)
            e = RegistrationHIP(p)
            ct, cs = e.upload(tgt), e.upload(src)
            guess = np.eye(4) if "dense" not in name else T @ synth.pose_matrix([0.1, 0.05, 0], [0, 0, 0.005])
            e.setInputTarget(ct)
            e.setInputSource(cs)
            e.align(guess)   # warm-up
            walls = []
            for rep in range(3):
                ct.invalidate(), cs.invalidate()
                e.profile_enable(True)
                e.profile_read(True)
                t0 = time.perf_counter()
                r = e.align(guess)
                fit = e.getFitnessScore()
                walls.append(time.perf_counter() - t0)
                prof = e.profile_read(True)
                e.profile_enable(False)
            t0 = time.perf_counter()
            r2 = e.align(guess)           # warm: index / covariances / voxels cached
            warm = time.perf_counter() - t0
            dt, dr = synth.pose_error(r.matrix(), T)
            print(f"{name} method {method}: n={len(src)} it={r.iterations} tries={r.lm_tries} conv={r.converged} cold wall {min(walls) * 1e3:.2f} ms (align+fitness) warm align {warm * 1e3:.2f} ms "
                  f"| vs GT {dt:.3f} m | stages ms(launches): " + " ".join(f"{s}={m:.3f}({c})" for s, (m, c) in prof.items() if c))
            e.close()


def cpu_scaling():
    tgt, src, T = synth.make_pair("HDL-64E", 0)
    for nt in (8, 32, 64, 128, os.cpu_count()):
        O.set_num_threads(nt)
        o = O.OracleRegistration(O.default_params(O.HGS_FAST_GICP))
        o.setInputTarget(tgt)
        o.setInputSource(src)
        o.align(np.eye(4))
        t0 = time.perf_counter()
        o.setInputSource(src.copy())
        t1 = time.perf_counter()
        r = o.align(np.eye(4))
        t2 = time.perf_counter()
        o.getFitnessScore()
        t3 = time.perf_counter()
        print(f"cpu oracle GICP 120k threads={nt}: set_source {t1 - t0:.3f}s align {t2 - t1:.3f}s (it {r.iterations}) fitness {t3 - t2:.3f}s")


def ndt_trace():
    """Device-side per-iteration trace (HGS_TRACE=1) next to the oracle's, for the VLP-16 case whose iteration diverges."""
    os.environ["HGS_TRACE"] = "1"
    tgt, src, T = synth.make_pair("VLP-16", 2, downsample=0.1)
    wild = T @ synth.pose_matrix([0.05, 0.02, 0.0], [0.0, 0.0, 0.004])
    p = O.default_params(O.HGS_NDT_OMP)
    p.resolution, p.max_iterations = 1.0, 3
    e, o = hip(p), O.OracleRegistration(p)
    for r in (e, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    ro = o.align(wild)
    np.set_printoptions(precision=9, suppress=True, linewidth=200)
    print("oracle trace (p, score, step):")
    print(o.trace())
    sys.stdout.flush()
    re = e.align(wild)
    e.synchronize()
    print("hip iterations", re.iterations, "oracle", ro.iterations, "diff", synth.pose_error(re.matrix(), ro.matrix()))
    e.close()
    del os.environ["HGS_TRACE"]


if __name__ == "__main__":
    which = sys.argv[1:] or ["ndt", "stages", "cpu"]
    if "trace" in which:
        ndt_trace()
    if "ndt" in which:
        ndt_parity()
    if "stages" in which:
        stage_times()
    if "cpu" in which:
        cpu_scaling()
