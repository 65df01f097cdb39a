#!/bin/bash
# Round 3 extras on the final library: config 4 on the default engine (SURVEY 8d asks for FAST_GICP and NDT), the metric with PLANE regularisation, smoke.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -c "
import ctypes; from hdl_graph_slam_amd import _lib as L; print('abi', L.lib().hgs_abi_version())"
timeout 900 python bench.py --config 4 --method NDT_OMP --seeds 1 --cpu-sample 2 2>/dev/null | grep '^{' > gpurun_out/r03_bench_config4_ndt.json; python -c "
import json; r=json.load(open('gpurun_out/r03_bench_config4_ndt.json')); print('config 4 NDT:', r['value'], r['ms_per_step'], r['mean_iterations'], r['converged'], r['roofline']['frac'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'])"
timeout 900 python bench.py --regularization PLANE --no-ndt-record --seeds 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_bench_metric_plane.json; python -c "
import json; r=json.load(open('gpurun_out/r03_bench_metric_plane.json')); print('metric PLANE:', r['value'], r['ms_per_step'], r['mean_iterations'], r['pose_rmse_vs_ground_truth'], r['roofline']['stage_ms_per_step'])"
