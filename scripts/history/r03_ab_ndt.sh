#!/bin/bash
# Same-box A/B of library variants on the NDT_OMP batch (ab_libs/<name>.so; VARIANTS="vH vN" by default).  Round 3's last visit: vH = the
# committed library, vN = staged tile front + scalar exp coefficients.  Prints the fields that must agree exactly between variants (the sums
# are order-independent: iterations, poses and inliers are bit-identical), then the NDT GPU tests on the LAST variant, then a second repetition.
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
V="${VARIANTS:-vH vN}"
LAST=$(echo $V | awk '{print $NF}')
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
{ for v in $V; do run $v; done; } 2>&1 | tee -a gpurun_out/r03_ab_ndt.log
cp ab_libs/$LAST.so hdl_graph_slam_amd/lib/libhgs_hip.so
timeout ${TEST_TIMEOUT:-120} python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ndt" -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/r03_ab_ndt.log
{ for v in $V; do run $v; done; } 2>&1 | tee -a gpurun_out/r03_ab_ndt.log   # second repetition if the visit's time allows
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
