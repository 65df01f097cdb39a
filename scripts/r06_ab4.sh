#!/bin/bash
# Round 6, fourth A/B: the prefilter on the voxel grid (engine option prefilter_fast=1: distance filter inside the voxel-grid kernels, RadiusOutlierRemoval by voxel key,
# one host read-back) against the separate passes + search tree (0) on the KITTI launch file's pipeline.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab4.log
timeout 1500 python -m pytest tests/test_prefilter.py tests/test_odometry.py tests/test_hip_parity.py tests/test_map_cloud.py -m gpu -x -q 2>&1 | tail -4 | tee -a $LOG
for rep in 1 2 3; do for f in 0 1; do
  export HGS_ENGINE_OPTIONS=prefilter_fast=$f
  echo -n "prefilter_fast=$f kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
done; done
unset HGS_ENGINE_OPTIONS
TAG=r06_fast SKIP_CFG2=1 bash scripts/r06_timelines.sh > /dev/null 2>&1
grep -A3 "^kernel time" gpurun_out/r06_fast_kitti_timeline.md | head -5 | tee -a $LOG
