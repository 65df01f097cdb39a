#!/bin/bash
# Round 6: the metric line over scene seeds 0-9 (mean +- std over the 10), three engines.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
for tag in "" plane ndt; do
  case "$tag" in
    "") M="--method FAST_GICP"; OUT=gpurun_out/r06_bench_metric_seeds10.json;;
    plane) M="--method FAST_GICP --regularization PLANE"; OUT=gpurun_out/r06_bench_metric_plane_seeds10.json;;
    ndt) M="--method NDT_OMP"; OUT=gpurun_out/r06_bench_metric_ndt_seeds10.json;;
  esac
  timeout 1500 python bench.py $M --seeds 10 --no-cpu-baseline --no-ndt-record --no-plane-record 2>/dev/null | grep '^{' | tail -1 > $OUT
  python -c "
import json; r = json.load(open('$OUT')); print('$tag seeds:', r['value'], r['value_by_scene_seed'], r['value_mean_std_over_seeds'], r['mean_iterations_by_scene_seed'])"
done
