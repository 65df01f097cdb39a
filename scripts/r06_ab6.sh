#!/bin/bash
# Round 6, sixth A/B: (a) prefilter — k_pf_grid_radius_flags with G lanes per centroid (one z-layer each) and k_pf_voxel_centroids reading its run four entries at a
# time (library `base` = before); (b) engine option warm_tree: small single-cloud launches of k_knn_cov / k_gicp_linearize touch the whole tree once per block first.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab6.log
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
echo -n "new library, prefilter + odometry + parity tests: " | tee -a $LOG
timeout 1500 python -m pytest tests/test_prefilter.py tests/test_odometry.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
echo -n "warm_tree=1, parity tests: " | tee -a $LOG
HGS_ENGINE_OPTIONS="warm_tree=1" timeout 1500 python -m pytest tests/test_odometry.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
for rep in 1 2 3; do for v in base new:0 new:1; do
  lib=${v%%:*}; warm=${v##*:}
  if [ "$lib" = base ]; then cp ab_libs/base.so hdl_graph_slam_amd/lib/libhgs_hip.so; unset HGS_ENGINE_OPTIONS
  else cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so; export HGS_ENGINE_OPTIONS="warm_tree=$warm"; fi
  echo -n "$v kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "$v cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
unset HGS_ENGINE_OPTIONS
TAG=r06_ab6 SKIP_CFG2=1 bash scripts/r06_timelines.sh > /dev/null 2>&1
HGS_ENGINE_OPTIONS="warm_tree=1" TAG=r06_ab6_warm SKIP_CFG2=1 bash scripts/r06_timelines.sh > /dev/null 2>&1
