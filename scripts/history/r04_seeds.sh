#!/bin/bash
# The metric's workload over scene seeds 0-9 (the timed region is seed 0's; the others are timed on a quarter of the steps): FAST_GICP, PLANE, NDT_OMP
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
timeout 900 python bench.py --method FAST_GICP --seeds 10 --no-cpu-baseline --no-ndt-record 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_metric_seeds10.json
timeout 900 python bench.py --method FAST_GICP --regularization PLANE --seeds 10 --no-cpu-baseline --no-ndt-record 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_metric_plane_seeds10.json
timeout 900 python bench.py --method NDT_OMP --seeds 10 --no-cpu-baseline --no-ndt-record 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_metric_ndt_seeds10.json
python - <<'PY'
import json
for n in ("metric", "metric_plane", "metric_ndt"):
    r = json.load(open(f"gpurun_out/r04_bench_{n}_seeds10.json"))
    print(n, r["value"], r["value_by_scene_seed"], r["value_mean_std_over_seeds"], r["mean_iterations_by_scene_seed"])
PY
