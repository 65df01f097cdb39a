#!/bin/bash
# CPU side of the A/B of scripts/pending: ab_libs/vH.so (the tree as it is), ab_libs/<patch>.so per patch, and mfma_all.so =
# ndt_pass_items32 + ndt_reduction_mfma + gicp_linearize_sums_mfma (ndt_reduction_scalar_scales and ndt_reduction_mfma exclude each other).
# GPU side (NDT):       gpurun -- 'VARIANTS="vH ndt_pass_items32 ndt_reduction_scalar_scales ndt_reduction_mfma mfma_all" bash scripts/r03_ab_ndt.sh'
# GPU side (FAST_GICP): gpurun -- 'VARIANTS="vH gicp_linearize_sums_mfma mfma_all" BENCH_FLAGS=" " bash scripts/r03_ab.sh'
set -eu
cd "$(dirname "$0")/../.."
git diff --quiet -- hdl_graph_slam_amd/csrc oracle || { echo "uncommitted changes in the patched directories"; exit 1; }
restore() { git checkout -- hdl_graph_slam_amd/csrc oracle; }
trap restore EXIT
scripts/build_variant.sh vH > /dev/null
for p in scripts/pending/*.patch; do
  n=$(basename "$p" .patch)
  git apply "$p"
  scripts/build_variant.sh "$n" > /dev/null || echo "build of $n failed"
  restore
done
git apply scripts/pending/ndt_pass_items32.patch scripts/pending/ndt_reduction_mfma.patch scripts/pending/gicp_linearize_sums_mfma.patch
scripts/build_variant.sh mfma_all > /dev/null || echo "build of mfma_all failed"
restore
ls -la ab_libs/
