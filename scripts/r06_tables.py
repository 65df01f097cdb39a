#!/usr/bin/env python3
"""Prints the numbers BASELINE.md §4 / DESIGN.md §6 quote for round 6 out of the committed bench lines (profiles/r06_bench_*.json)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda name: os.path.join(ROOT, "profiles", name)


def load(name):
    try:
        return json.load(open(P(name)))
    except (OSError, ValueError):
        return None


def engine(tag, r):
    c, ro = r.get("cpu_baseline") or {}, r.get("roofline") or {}
    print(f"{tag}: {r['value']} reg/s, {r['ms_per_step']} ms/step, {r.get('mean_iterations')} its, conv {r.get('converged')}, best {r.get('best_candidate')}")
    print(f"   cpu {c.get('value_over_all_candidates_by_variant') or c.get('value')} cores {c.get('cores')} gpu/cpu {c.get('gpu_over_cpu')} checked {c.get('candidates_checked')} argmin {c.get('oracle_argmin_agrees')} "
          f"its_equal {c.get('iterations_equal')} pose {c.get('max_pose_diff_vs_gpu_m')} fit {c.get('max_fitness_rel_diff_vs_gpu')}")
    print(f"   roofline {ro.get('bound')} {ro.get('kernel')} achieved {ro.get('achieved')} {ro.get('unit')} frac {ro.get('frac')} avg_launch_us {ro.get('avg_launch_us')} launches {ro.get('launches')} "
          f"alg_bytes {ro.get('algorithmic_bytes_per_launch')} traffic {ro.get('traffic')} counters_frac {ro.get('hbm_frac_from_counters')} valu {ro.get('valu')}")
    print(f"   stages {ro.get('stage_ms_per_step')} profiled_step {ro.get('profiled_step_ms')}")


d = load("r06_bench_driver_cmd.json")
if d:
    engine("FAST_GICP FROBENIUS", d)
    engine("FAST_GICP PLANE", d["fast_gicp_plane"])
    engine("NDT_OMP", d["ndt_omp"])
    print("r02 set", d.get("r02_candidate_set"))
c2 = load("r06_bench_config2.json")
if c2:
    print("config 2:", c2["value"], "reg/s", c2["step_ms"], "warm", c2.get("warm_align_ms"), "its", c2.get("iterations"), "cpu", (c2.get("cpu_baseline") or {}).get("value"), c2["roofline"].get("stage_ms_per_step"))
c3 = load("r06_bench_config3.json")
if c3:
    t = c3["trajectory_error_vs_ground_truth"]
    print("config 3:", c3["config"]["workload"][:90], "| value", c3["value"], "p50", c3["latency_ms"]["p50"], "its", c3["mean_iterations"], "track", t, "cpu", (c3.get("cpu_baseline") or {}).get("value"))
    for k in c3:
        if k.startswith("at_"):
            print("  ", k, c3[k]["value"], c3[k]["latency_ms"]["p50"], c3[k]["mean_iterations"], c3[k]["trajectory_error_vs_ground_truth"])
    for k in ("kitti_prefilter_ndt_omp", "kitti_launch_fast_gicp"):
        if k in c3:
            print("  ", k, c3[k]["value"], "p50", c3[k]["latency_ms"]["p50"], "its", c3[k]["mean_iterations"], c3[k]["trajectory_error_vs_ground_truth"], "pts", c3[k]["points_after_prefilter"], "cpu",
                  (c3[k].get("cpu_baseline") or {}).get("value"), "oracle", c3[k].get("oracle_stream"))
    a = c3.get("adapter_path") or {}
    for k in ("raw_sweeps", "behind_voxelgrid_0_25"):
        if k in a:
            v = a[k]
            print("   adapter", k, {m: v[m]["p50_ms"] for m in v if isinstance(v[m], dict) and "p50_ms" in v[m]}, "ratio", v.get("adapter_over_c_abi_p50"))
for name in ("r06_bench_config4.json", "r06_bench_config4_plane.json", "r06_bench_config4_ndt.json", "r06_bench_config5.json", "r06_bench_world1_rccl.json", "r06_bench_single_process.json"):
    r = load(name)
    if r:
        c = r.get("cpu_baseline") or {}
        print(name, r["value"], r["ms_per_step"], r.get("step_ms"), "its", r.get("mean_iterations", r.get("iterations")), "cpu", c.get("value"), "checked", c.get("candidates_checked"), c.get("oracle_argmin_agrees"),
              (r.get("config") or {}).get("exchange"), r.get("warm_align_ms"))
