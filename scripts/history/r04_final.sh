#!/bin/bash
# Round 4 final visit on the final library: the -m gpu suite, the driver's bench command, profiles of the three lines, the other configurations
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r04_gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd.log 2>gpurun_out/r04_bench_driver_cmd.err; echo "bench exit $?"
grep '^{' gpurun_out/r04_bench_driver_cmd.log | tail -1 > gpurun_out/r04_bench_driver_cmd.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_bench_driver_cmd.json"))
def brief(x):
    return {k: x.get(k) for k in ("value", "ms_per_step", "mean_iterations", "converged", "best_candidate")} | {"rmse": x["pose_rmse_vs_ground_truth"], "cpu": {k: x["cpu_baseline"].get(k) for k in ("value", "cores", "candidates_checked", "oracle_argmin_agrees", "max_pose_diff_vs_gpu_m", "iterations_equal")} if x.get("cpu_baseline") else None, "roof": {k: x["roofline"].get(k) for k in ("bound", "kernel", "achieved", "frac", "avg_launch_us", "hbm_frac_from_counters", "profiled_step_ms", "stage_ms_per_step")}}
print("FAST_GICP", brief(r)); print("PLANE", brief(r["fast_gicp_plane"])); print("NDT", brief(r["ndt_omp"])); print("r02 set", r["r02_candidate_set"])
PY
bash scripts/r04_profile.sh
bash scripts/r04_configs.sh
