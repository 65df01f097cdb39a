#!/usr/bin/env python3
"""NDT: deviation of the order-independent exact sum (the device's definition; oracle sum mode 1) from ndt_omp's serial double sum
(oracle sum mode 0), measured run for run to convergence on the HDL-64E 0.2 m-voxel pair, 100 guesses per neighbourhood mode.
CPU only (both sides are the oracle); the `-m gpu` test tests/test_hip_parity.py::test_ndt_deviation_from_the_serial_sum makes the
same measurement with the HIP engine on the exact side.  Prints a markdown table (profiles/r03_ndt_sum_mode_deviation.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O  # noqa: E402
import parity_checks as PC  # noqa: E402
from hdl_graph_slam_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seeds = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
    O.set_num_threads(min(32, os.cpu_count() or 1))
    rows = []
    for seed in seeds:
        tgt, src, T = synth.make_pair("HDL-64E", seed, downsample=0.2)
        for r in PC.ndt_serial_sum_deviation(lambda p: PC.make_oracle(p).set_ndt_sum_mode(1), tgt, src, T, n_guesses=n, seed=seed):
            r["scene_seed"], r["points"] = seed, int(len(src))
            rows.append(r)
            print(json.dumps(r), flush=True)
    print("\n| scene seed | mode | guesses | > 1e-3 m / rad | other iteration count | max dt [m] | p99 dt [m] | median dt [m] |\n|---|---|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        print(f"| {r['scene_seed']} | {r['mode']} | {r['guesses']} | {r['beyond_1e-3']} | {r['other_iteration_count']} | {r['max_dt_m']:.2e} | {r['p99_dt_m']:.2e} | {r['median_dt_m']:.2e} |")


if __name__ == "__main__":
    main()
