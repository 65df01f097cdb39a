#!/bin/bash
# Round 4, second GPU visit: the whole -m gpu suite on the merged library (LDS lane-slot reduction, hardened sharded entry point, all-64 oracle
# tests), the driver's bench command, and kernel timelines of ONE registration (configs 2 and 3) for the latency path.
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r04_gputest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd.log 2>gpurun_out/r04_bench_driver_cmd.err; echo "bench exit $?"
grep '^{' gpurun_out/r04_bench_driver_cmd.log | tail -1 > gpurun_out/r04_bench_driver_cmd.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_bench_driver_cmd.json"))
def brief(x):
    return {k: x.get(k) for k in ("value", "ms_per_step", "mean_iterations", "converged", "best_candidate")} | {"rmse": x["pose_rmse_vs_ground_truth"], "cpu": {k: x["cpu_baseline"].get(k) for k in ("value", "cores", "candidates_checked", "oracle_argmin_agrees", "max_pose_diff_vs_gpu_m", "iterations_equal")} if x.get("cpu_baseline") else None, "roof": {k: x["roofline"].get(k) for k in ("bound", "kernel", "achieved", "frac", "avg_launch_us", "hbm_frac_from_counters", "profiled_step_ms", "stage_ms_per_step")}}
print("FAST_GICP", brief(r)); print("PLANE", brief(r["fast_gicp_plane"])); print("NDT", brief(r["ndt_omp"])); print("r02 set", r["r02_candidate_set"])
PY
for cfg in 2 3; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/trace_cfg$cfg" -o t -- python "$ROOT/bench.py" --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0 > "$ROOT/gpurun_out/trace_cfg$cfg.log" 2>&1); echo "trace cfg $cfg exit $?"
  f=$(find gpurun_out/trace_cfg$cfg -name "*kernel_trace.csv" | head -1)
  g=$(find gpurun_out/trace_cfg$cfg -name "*kernel_stats.csv" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0"; echo; echo '```'; grep '^{' gpurun_out/trace_cfg$cfg.log | tail -1 | cut -c1-1500; echo '```'; echo;
    [ -n "$g" ] && python scripts/prof_summary.py "$g"; echo; [ -n "$f" ] && python scripts/trace_timeline.py "$f"; } > gpurun_out/r04_config${cfg}_timeline.md
  find gpurun_out/trace_cfg$cfg -name "*.csv" -size +8M -delete
done
head -60 gpurun_out/r04_config2_timeline.md | cut -c1-200
