#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
VARIANTS="v5 v6 v7" REPS=2 bash scripts/r03_ab.sh
cp ab_libs/v6.so hdl_graph_slam_amd/lib/libhgs_hip.so
echo "== v6 unfused (HGS_GICP_FUSED=0)"; HGS_GICP_FUSED=0 VARIANTS="v6" REPS=1 bash scripts/r03_ab.sh
echo "== v6 8d set"; BENCH_FLAGS=" " VARIANTS="v5 v6" REPS=1 bash scripts/r03_ab.sh
echo "== NDT (svd in registers): v5 vs v6"; VARIANTS="v5 v6" METHODS=NDT_OMP REPS=1 bash scripts/r03_ab.sh
CONFIGS="2 3" FLAGS="--no-cpu-baseline --seeds 1 --oracle-sweeps 0" bash scripts/r03_configs.sh
