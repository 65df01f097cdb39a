#!/bin/bash
# Round 5: the metric's three lines over scene seeds 0-9 (the timed region is seed 0's; the others are reported next to it)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 1200 python bench.py --steps 20 --warmup 5 --seeds 10 --no-cpu-baseline --no-plane-record --no-ndt-record 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r05_bench_metric_seeds10.json
timeout 600 python bench.py --steps 20 --warmup 5 --seeds 10 --no-cpu-baseline --regularization PLANE --no-ndt-record 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r05_bench_metric_plane_seeds10.json
timeout 900 python bench.py --steps 8 --warmup 3 --seeds 10 --no-cpu-baseline --method NDT_OMP 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r05_bench_metric_ndt_seeds10.json
python - <<'PY'
import json
for f in ("metric", "metric_plane", "metric_ndt"):
    r = json.load(open(f"gpurun_out/r05_bench_{f}_seeds10.json"))
    print(f, r["value"], r["value_mean_std_over_seeds"], r["value_by_scene_seed"], r["mean_iterations_by_scene_seed"])
PY
