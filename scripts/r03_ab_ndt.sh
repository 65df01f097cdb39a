#!/bin/bash
# Last same-box A/B of the round (NDT_OMP): vH = the committed library, vN = staged tile front + scalar exp coefficients.  Prints the
# fields that must agree exactly between the two (the sums are order-independent: iterations, poses and inliers are bit-identical).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
run() {
  v=$1
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  echo -n "$v NDT_OMP: "
  timeout 100 python bench.py --method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1 --distinct 4 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], 'its', r['mean_iterations'], 'conv', r['converged'], 'rmse', r['pose_rmse_vs_ground_truth']['translation_m'], r['pose_rmse_vs_ground_truth']['rotation_rad'], 'best', r['best_candidate'], 'inl', r['num_inliers_mean'], 'pass us', r['roofline']['avg_launch_us'])
"
}
{ run vH; run vN; } 2>&1 | tee -a gpurun_out/r03_ab_ndt.log
cp ab_libs/vN.so hdl_graph_slam_amd/lib/libhgs_hip.so
timeout 60 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ndt" -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/r03_ab_ndt.log
{ run vH; run vN; } 2>&1 | tee -a gpurun_out/r03_ab_ndt.log   # second repetition if the visit's time allows
