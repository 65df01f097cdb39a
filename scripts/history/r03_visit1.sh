#!/bin/bash
# Round 3, visit 1: GPU parity suite on the quad-walk library, then same-box A/B of library variants (ab_libs/*.so).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
echo "== pytest -m gpu (current library)"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 --timeout-method=thread -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/r03_v1_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03_v1_pytest_gpu.log
echo "== A/B"
VARIANTS="${VARIANTS:-v0 v1 v2 v3}" METHODS="${METHODS:-FAST_GICP}" bash scripts/r02_ab_libs.sh 2>&1 | tee gpurun_out/r03_ab1.log
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
