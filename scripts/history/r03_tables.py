#!/usr/bin/env python3
"""Builds the round-3 results table of BASELINE.md §4 from the committed bench lines (profiles/r03_bench_*.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "profiles")


def load(name):
    p = os.path.join(D, name)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        return None
    with open(p) as fh:
        return json.load(fh)


def cpu(r):
    c = r.get("cpu_baseline") or {}
    one = (c.get("single_thread") or {}).get("value")
    return (f"{one:.3g} reg/s" if one else "—"), (f"{c['value']:.3g} reg/s @ {c['cores']} thr" if c else "—"), c.get("value")


def seeds(r):
    v = r.get("value_by_scene_seed") or []
    ms = r.get("value_mean_std_over_seeds") or [0, 0]
    return f"seeds 0–{len(v) - 1}: {ms[0]:.0f} ± {ms[1]:.0f}" if len(v) > 1 else ""


def roof(r):
    f = r["roofline"]
    return f"`{f['kernel']}`: {f['achieved']:.0f} GB/s ({100 * f['frac_of_measured_copy_ceiling']:.1f} % / {100 * f['frac']:.1f} %)"


rows = []
m = load("r03_bench_metric_seeds10.json") or load("r03_bench_driver_cmd.json")
if m:
    one, n, v = cpu(m)
    rows.append(("metric: 64 × HDL-64E ~119 k candidates / GPU, SURVEY 8d candidate set", "FAST_GICP + getFitnessScore, cold", one, n,
                 f"**{m['value']:.0f} reg/s** ({m['ms_per_step']:.2f} ms per 64-candidate step, {m['mean_iterations']:.2f} LM iterations); {seeds(m)}; keyframes resident {m['resident_keyframes_value']:.0f}"
                 + (f"; rounds 1–2 candidate set: {m['r02_candidate_set']['value']:.0f}" if m.get("r02_candidate_set") else ""),
                 f"{m['value'] / v:.0f}×" if v else "—", f"{m['cpu_baseline']['max_pose_diff_vs_gpu_m']:.1e} / {m['cpu_baseline']['max_pose_diff_vs_gpu_rad']:.1e}" if m.get("cpu_baseline") else "—", roof(m)))
nd = load("r03_bench_metric_ndt_seeds10.json")
if nd:
    one, n, v = cpu(nd)
    rows.append(("metric workload, default engine", f"NDT_OMP DIRECT7 res 1.0 + getFitnessScore ({nd['mean_iterations']:.0f} iterations per candidate)", one, n,
                 f"**{nd['value']:.0f} reg/s** ({nd['ms_per_step']:.1f} ms per step); {seeds(nd)}", f"{nd['value'] / v:.0f}×" if v else "—",
                 f"{nd['cpu_baseline']['max_pose_diff_vs_gpu_m']:.1e} / {nd['cpu_baseline']['max_pose_diff_vs_gpu_rad']:.1e}" if nd.get("cpu_baseline") else "—", roof(nd)))
c2 = load("r03_bench_config2.json")
if c2:
    one, n, v = cpu(c2)
    rows.append(("2. HDL-32E ~65 k pair", f"FAST_GICP, single `align` from the identity guess ({c2['iterations']} iterations)", one, n + " (cold)",
                 f"cold {c2['step_ms']['p50']:.3f} ms p50 = **{c2['value']:.0f} reg/s**; warm (target cached) {c2['warm_align_ms']['p50']:.3f} ms; {seeds(c2)}", f"{c2['value'] / v:.0f}×" if v else "—",
                 f"{c2['cpu_baseline']['max_pose_diff_vs_gpu_m']:.1e} / {c2['cpu_baseline']['max_pose_diff_vs_gpu_rad']:.1e}" if c2.get("cpu_baseline") else "—", roof(c2)))
c3 = load("r03_bench_config3.json")
if c3:
    one, n, v = cpu(c3)
    a3 = c3.get("at_3_mps") or {}
    te = c3["trajectory_error_vs_ground_truth"]
    rows.append(("3. 64-beam ~118 k stream, 8 m/s (SURVEY 8d)", "NDT_OMP DIRECT7 res 1.0, KITTI keyframe rule, **H2D included**", one, n,
                 f"**loses track** (final error {te['final_translation_m']:.1f} m after {c3['steps']} sweeps, {c3['mean_iterations']:.1f} iterations per sweep: the basin at resolution 1.0 is narrower than 0.8 m per sweep on this scene; "
                 f"the CPU oracle through the same caller ends {c3['oracle_stream']['max_translation_diff_vs_device_m']:.0e} m from the device after {c3['oracle_stream']['sweeps']} sweeps, i.e. it does the same); latency p50 {c3['latency_ms']['p50']:.2f} ms",
                 f"{c3['value'] / v:.0f}×" if v else "—", "identical trajectory (see left)", roof(c3)))
    if a3:
        rows.append(("3. same stream at 3 m/s (rounds 1–2)", "same", "—", "—",
                     f"latency p50 {a3['latency_ms']['p50']:.3f} / p90 {a3['latency_ms']['p90']:.3f} / p99 {a3['latency_ms']['p99']:.3f} ms at {a3['mean_iterations']:.1f} iterations mean (max {a3['max_iterations']}) = "
                     f"{1e3 * a3['latency_ms']['p50'] / a3['mean_iterations']:.0f} µs per iteration; **{a3['value']:.0f} reg/s**; final error {a3['trajectory_error_vs_ground_truth']['final_translation_m']:.2f} m; oracle Δ "
                     f"{(a3.get('oracle_stream') or {}).get('max_translation_diff_vs_device_m', float('nan')):.0e} m", "—", "bit-identical stream (50-sweep test)", "—"))
c4 = load("r03_bench_config4.json")
if c4:
    one, n, v = cpu(c4)
    f4 = c4.get("fitness_score_max_range_4") or {}
    rows.append(("4. 1 × 512 candidates (HDL-32E, ~65 k pts), SURVEY 8d set", "FAST_GICP batch, all 512 on ONE GPU", one, n,
                 f"{c4['ms_per_step']:.1f} ms per 512-candidate detection = **{c4['value']:.0f} reg/s** ({c4['mean_iterations']:.2f} LM iterations); {seeds(c4)}; keyframes resident {c4['resident_keyframes_value']:.0f}; "
                 f"`fitness_score_max_range` 4.0: {f4.get('value', float('nan')):.0f} reg/s; 2 / 4 / 8 GPUs: the driver's SCALE run", f"{c4['value'] / v:.0f}×" if v else "—",
                 f"{c4['cpu_baseline']['max_pose_diff_vs_gpu_m']:.1e} / {c4['cpu_baseline']['max_pose_diff_vs_gpu_rad']:.1e}" if c4.get("cpu_baseline") else "—", roof(c4)))
c5 = load("r03_bench_config5.json")
if c5:
    one, n, v = cpu(c5)
    st = c5["roofline"]["stages"]
    rows.append(("5. 1 M-point synthetic", f"FAST_GICP, max_corr 1.0, single `align` ({c5['iterations']} iterations)", one, n,
                 f"cold {c5['step_ms']['p50']:.3f} ms, warm {c5['warm_align_ms']['p50']:.3f} ms; per stage (ms per align / % of 8 TB/s): " +
                 ", ".join(f"{k} {st[k]['ms_per_align']:.2f} / {100 * st[k]['frac_of_8TBps']:.1f} %" for k in ("covariance", "linearize", "error") if k in st) + f"; {seeds(c5)}",
                 f"{c5['value'] / v:.0f}×" if v else "—", f"{c5['cpu_baseline']['max_pose_diff_vs_gpu_m']:.1e} / {c5['cpu_baseline']['max_pose_diff_vs_gpu_rad']:.1e}" if c5.get("cpu_baseline") else "—", roof(c5)))
print("| config (BASELINE.json) | engine | CPU oracle 1 thr | CPU oracle N thr | 1× MI355X | speed-up vs N thr | pose Δ vs oracle (m / rad) | achieved algorithmic GB/s of the dominant kernel (% of 6.29 TB/s measured, % of 8 TB/s spec) |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
