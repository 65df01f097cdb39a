#!/bin/bash
# First GPU visit of a round: everything that was added without a GPU gets its hardware run, then the headline numbers.
#   gpurun --timeout 900 -- 'bash scripts/gpu_first_visit.sh'        (logs in gpurun_out/)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== pytest -m gpu"; timeout 600 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
echo "== bench (default)"; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-400
for ls in "" "--ndt-line-search"; do
  echo "== bench NDT_OMP $ls"; timeout 300 python bench.py --method NDT_OMP --steps 5 --warmup 1 --no-cpu-baseline $ls 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print(r['value'], r['ms_per_step'], r['mean_iterations'], r.get('mean_linearizations'), r['pose_rmse_vs_ground_truth'])" | tee -a gpurun_out/ndt_line_search.log
  echo "== odometry HDL-64E NDT 3 m/s $ls"; timeout 300 python scripts/odometry_stream.py --sensor HDL-64E --method NDT_OMP --scans 30 --oracle-scans 2 --speed 3 $ls 2>/dev/null | tail -1 | cut -c1-600 | tee -a gpurun_out/ndt_line_search.log
done
echo "== odometry HDL-64E NDT 8 m/s with the KITTI prefilter (voxel 0.25, SURVEY 8d cfg 3)"; timeout 300 python scripts/odometry_stream.py --sensor HDL-64E --method NDT_OMP --scans 30 --oracle-scans 2 --downsample 0.25 2>/dev/null | tail -1 | cut -c1-600 | tee -a gpurun_out/odometry_prefilter.log
echo "== rocprofv3 (whole-device launches)"
(cd /tmp && HGS_BATCH_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_lanes1" -o bench -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_lanes1.log" 2>&1); echo "prof exit $?"
f=$(find gpurun_out/prof_lanes1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python scripts/prof_summary.py "$f" | head -12
