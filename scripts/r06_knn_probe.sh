#!/bin/bash
# Round 6: per-wave phase clocks / step counts of small k_knn_cov launches (measurement build ab_libs/knnprobe.so = scripts/build_variant.sh knnprobe -DHGS_KNN_PROBE)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
cp ab_libs/knnprobe.so hdl_graph_slam_amd/lib/libhgs_hip.so
timeout 600 python scripts/probes/knn_probe.py 2>&1 | tee gpurun_out/r06_knn_probe.log
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
