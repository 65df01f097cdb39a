#!/bin/bash
# Round 6 final visit on the final library: the -m gpu suite, smoke, the driver's bench command, profiles of the three lines, the other configurations
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12 | grep -E "passed|failed|error" | tee gpurun_out/r06_gputest_final.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/r06_gputest_final.log
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.log 2>gpurun_out/r06_bench_driver_cmd.err; echo "bench exit $?"
grep '^{' gpurun_out/r06_bench_driver_cmd.log | tail -1 > gpurun_out/r06_bench_driver_cmd.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_bench_driver_cmd.json"))
def brief(x):
    c = x.get("cpu_baseline")
    return {k: x.get(k) for k in ("value", "ms_per_step", "mean_iterations", "converged", "best_candidate")} | {"cpu": {k: c.get(k) for k in ("value", "cores", "variants", "value_over_all_candidates_by_variant", "gpu_over_cpu", "candidates_checked", "oracle_argmin_agrees", "max_pose_diff_vs_gpu_m", "iterations_equal")} if c else None, "roof": {k: x["roofline"].get(k) for k in ("bound", "kernel", "achieved", "frac", "avg_launch_us", "hbm_frac_from_counters", "profiled_step_ms", "stage_ms_per_step")}}
print("FAST_GICP", brief(r)); print("PLANE", brief(r["fast_gicp_plane"])); print("NDT", brief(r["ndt_omp"])); print("r02 set", r["r02_candidate_set"])
PY
if [ -z "${SKIP_PROFILE:-}" ]; then bash scripts/r06_profile.sh; fi
if [ -z "${SKIP_CONFIGS:-}" ]; then bash scripts/r06_configs.sh; fi
