#!/usr/bin/env python3
"""Host-side parity sweep: the device functions (executed on the CPU by tests/emul) against the oracle over many scenes,
sensors, resolutions and adversarial clouds (duplicates, collinear runs, coincident planes, far outliers, tiny clouds).
Not part of the test suite (minutes of CPU time); prints one line per case and a summary, exits non-zero on a mismatch.

  python scripts/fuzz_parity_host.py [--seeds 6] [--quick] [--backend emul|simt]

--backend emul (default): the per-item device functions run by tests/emul/emul.cpp;  --backend simt: the product kernels
and engine themselves, compiled for the host against the SIMT emulation shim (tests/emul/simt.py) — slower, but it is the
code that runs on the GPU (packet walks, tile reductions, state machines, hash tables) that meets the adversarial inputs."""
import argparse
import os
import sys
import time
import traceback
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O  # noqa: E402
import parity_checks as PC  # noqa: E402
from emul import emul  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402


def adversarial(seed: int, base: np.ndarray) -> np.ndarray:
    """base cloud + exact duplicates, a collinear run per cell row, two coincident planes, far outliers."""
    rng = np.random.default_rng(seed)
    xyz = synth.xyz_of(base).copy()
    dup = xyz[rng.choice(len(xyz), len(xyz) // 20, replace=False)]
    t = np.linspace(0, 1, 40, dtype=np.float32)[:, None]
    lines = np.concatenate([np.float32([x0, y0, -1.0]) + t * np.float32([0.0, 0.0, 2.5]) for x0, y0 in rng.uniform(-20, 20, (30, 2))])
    g = np.stack(np.meshgrid(np.arange(-5, 5, 0.25, dtype=np.float32), np.arange(-5, 5, 0.25, dtype=np.float32)), -1).reshape(-1, 2)
    plane = np.concatenate([g, np.full((len(g), 1), 0.5, np.float32)], 1)
    far = rng.uniform(-1, 1, (8, 3)).astype(np.float32) * np.float32([400, 400, 30])
    out = np.concatenate([xyz, dup, lines.astype(np.float32), plane, plane, far])
    return synth.to_xyzi(out[rng.permutation(len(out))])


def cases(n_seeds: int, quick: bool, seed_base: int = 10):
    sensors = [("VLP-16", 0.1), ("VLP-16", None), ("HDL-32E", 0.25)] + ([] if quick else [("HDL-32E", None), ("HDL-64E", 0.2)])
    for seed in range(seed_base, seed_base + n_seeds):
        for sensor, ds in sensors:
            tgt, src, T = synth.make_pair(sensor, seed, downsample=ds)
            yield f"{sensor} seed {seed} ds {ds}", tgt, src, T
        tgt, src, T = synth.make_pair("VLP-16", seed, downsample=0.1)
        yield f"adversarial seed {seed}", adversarial(seed, tgt), adversarial(seed + 1000, src), T
    rng = np.random.default_rng(1)
    for n in (1, 7, 21, 64, 65, 300):  # 2 points: rank-1 covariances whose regularisation (arbitrary null-space basis) is ill-defined
        a = synth.to_xyzi(rng.normal(0, 3, (n, 3)).astype(np.float32))
        b = synth.to_xyzi(rng.normal(0, 3, (max(n, 25), 3)).astype(np.float32))
        yield f"tiny source {n}", b, a, np.eye(4)


def make_engine(params):
    if BACKEND == "emul":
        return emul.EmulRegistration(params)
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    p = L.HgsParams()
    for name, _ in L.HgsParams._fields_:
        setattr(p, name, getattr(params, name))
    return RegistrationHIP(p)


BACKEND = "emul"
GUESS_SALT = 0
SERIAL_SUM_DEVIATION = []   # (case, resolution, search, dt, dr) of the 3-iteration NDT prefixes against the oracle's serial-sum mode


def run_case(name, tgt, src, T, quick):
    rng = np.random.default_rng(zlib.crc32(name.encode()) + GUESS_SALT)   # (str hash() is salted per process: not reproducible)
    near = T @ synth.pose_matrix(rng.normal(0, 0.1, 3), rng.normal(0, 0.01, 3))
    done = []
    # ---- exact search + covariances + GICP
    p = O.default_params(O.HGS_FAST_GICP)
    e, o = make_engine(p), O.OracleRegistration(p)
    PC.load_pair(e, o, tgt, src)
    q = synth.xyz_of(src)
    PC.check_nn(e, o, q[:: max(1, len(q) // 4000)])
    done.append("nn")
    if len(tgt) > 20:
        PC.check_covariances(e, tgt)
        PC.check_gicp_linearize(e, o, near)
        PC.check_align(e, o, near, tol_m=1e-6, tol_rad=2e-5)
        done.append("gicp")
    # ---- VGICP
    if len(tgt) > 20:
        for res, search in ((1.0, O.HGS_DIRECT1), (0.5, O.HGS_DIRECT7)) + (() if quick else ((2.0, O.HGS_DIRECT27),)):
            p = O.default_params(O.HGS_FAST_VGICP)
            p.resolution, p.neighbor_search = res, search
            e, o = make_engine(p), O.OracleRegistration(p)
            PC.load_pair(e, o, tgt, src)
            PC.check_gicp_linearize(e, o, near)
        done.append("vgicp")
    # ---- NDT cells + derivatives (+ short prefixes of align)
    for res, search in ((1.0, O.HGS_DIRECT7), (0.5, O.HGS_DIRECT1), (2.0, O.HGS_KDTREE)):
        p = O.default_params(O.HGS_NDT_OMP)
        p.resolution, p.neighbor_search, p.max_iterations = res, search, 3
        # the product sums the per-point terms exactly (order-independent); the oracle's exact-sum mode (1) is the like-for-like
        # partner and must agree to 1e-6.  The upstream-faithful serial sum (mode 0; ndt_omp reduces serially in index order) differs
        # from it in the last bits of every pass, which a nearly singular Newton system (DIRECT1 at 0.5 m) can turn into
        # 1e-5 .. 1e-3 m within the three iterations of this prefix on some guesses: BOTH comparisons are reported, the second
        # as a number (the to-convergence measurement is scripts/ndt_sum_mode_deviation.py and the -m gpu test of the same name).
        e, o = make_engine(p), O.OracleRegistration(p).set_ndt_sum_mode(1)
        o_serial = O.OracleRegistration(p)
        PC.load_pair(e, o, tgt, src)
        if len(o.ndt_cells()[0]) == 0:
            continue
        PC.check_ndt_cells(e, o)
        for _ in range(2):
            p6 = np.concatenate([near[:3, 3] + rng.normal(0, 0.05, 3), rng.normal(0, 0.02, 3)])
            so = o.ndt_derivatives(p6)[0]
            if abs(so) > 0:
                PC.check_ndt_derivatives(e, o, p6)
        PC.check_align(e, o, near, tol_m=1e-6, tol_rad=1e-6)
        o_serial.setInputTarget(tgt)
        o_serial.setInputSource(src)
        dt, dr = synth.pose_error(e.align(near).matrix(), o_serial.align(near).matrix())
        SERIAL_SUM_DEVIATION.append((name, res, int(search), float(dt), float(dr)))
    done.append("ndt")
    return done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--seed-base", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--backend", default="emul", choices=["emul", "simt"])
    ap.add_argument("--guess-salt", type=int, default=0, help="added to the per-case seed of the random guesses (another draw of guesses)")
    args = ap.parse_args()
    global BACKEND, GUESS_SALT
    BACKEND, GUESS_SALT = args.backend, args.guess_salt
    if BACKEND == "simt":
        from emul import simt
        from hdl_graph_slam_amd import _lib as L
        L.LIB_PATH, L._lib = simt.build(), None
    failures = 0
    t0 = time.time()
    for name, tgt, src, T in cases(args.seeds, args.quick, args.seed_base):
        t1 = time.time()
        try:
            done = run_case(name, tgt, src, T, args.quick)
            print(f"ok   {name:32s} n={len(tgt):6d}/{len(src):6d} {'+'.join(done)} {time.time() - t1:.1f}s", flush=True)
        except Exception:
            failures += 1
            print(f"FAIL {name}", flush=True)
            traceback.print_exc()
    if SERIAL_SUM_DEVIATION:
        dev = SERIAL_SUM_DEVIATION
        beyond = [d for d in dev if d[3] > 1e-3 or d[4] > 1e-3]
        print(f"NDT 3-iteration prefixes vs the SERIAL sum (ndt_omp's association, oracle mode 0): {len(dev)} runs, max {max(d[3] for d in dev):.2e} m / "
              f"{max(d[4] for d in dev):.2e} rad, {len(beyond)} beyond 1e-3: {[(d[0], d[1], d[2]) for d in beyond]}")
    print(f"{failures} failing case(s), {time.time() - t0:.0f}s")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
