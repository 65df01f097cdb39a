#!/bin/bash
# Round 3 measurement visit on the final code: profiles (trace + PMC, both engines), the driver's command, every BASELINE configuration
# over scene seeds 0-9 with the one-thread CPU column.  Outputs under gpurun_out/ -> copied into profiles/ afterwards.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
METHODS="FAST_GICP NDT_OMP" bash scripts/r03_profile.sh
echo "== driver command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03_bench_driver_cmd.err | grep '^{' > gpurun_out/r03_bench_driver_cmd.json; python -c "
import json; r=json.load(open('gpurun_out/r03_bench_driver_cmd.json')); print('driver cmd:', r['value'], r['ms_per_step'], r['mean_iterations'], r['roofline']['frac'], r['roofline']['kernel'], r['roofline']['avg_launch_us'], r['roofline']['traffic'], '| ndt', r['ndt_omp']['value'], r['ndt_omp']['roofline']['frac'], '| r02 set', r['r02_candidate_set']['value'], '| cpu', r['cpu_baseline']['value'])"
echo "== metric over seeds 0-9"
timeout 900 python bench.py --seeds 10 --cpu-single-thread 2>/dev/null | grep '^{' > gpurun_out/r03_bench_metric_seeds10.json; python -c "
import json; r=json.load(open('gpurun_out/r03_bench_metric_seeds10.json')); print('metric seeds:', r['value_by_scene_seed'], r['value_mean_std_over_seeds'], r['mean_iterations_by_scene_seed'], r['cpu_baseline']['value'], r['cpu_baseline'].get('single_thread'))"
timeout 900 python bench.py --method NDT_OMP --seeds 10 --cpu-single-thread 2>/dev/null | grep '^{' > gpurun_out/r03_bench_metric_ndt_seeds10.json; python -c "
import json; r=json.load(open('gpurun_out/r03_bench_metric_ndt_seeds10.json')); print('ndt seeds:', r['value'], r['value_by_scene_seed'], r['value_mean_std_over_seeds'], r['cpu_baseline']['value'], r['cpu_baseline'].get('single_thread'))"
CONFIGS="${CONFIGS:-2 3 4 5}" FLAGS="--seeds 10 --cpu-single-thread" bash scripts/r03_configs.sh
