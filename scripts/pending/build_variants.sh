#!/bin/bash
# CPU side of the A/B of scripts/pending: ab_libs/vH.so (the tree as it is), ab_libs/<patch>.so per patch, ab_libs/all.so with every patch.
# GPU side:  gpurun -- 'VARIANTS="vH ndt_pass_items32 ndt_reduction_scalar_scales all" bash scripts/r03_ab_ndt.sh'
set -eu
cd "$(dirname "$0")/../.."
git diff --quiet -- hdl_graph_slam_amd/csrc || { echo "csrc has uncommitted changes"; exit 1; }
scripts/build_variant.sh vH > /dev/null
for p in scripts/pending/*.patch; do
  n=$(basename "$p" .patch)
  git apply "$p"
  scripts/build_variant.sh "$n" > /dev/null || echo "build of $n failed"
  git apply -R "$p"
done
git apply scripts/pending/*.patch
scripts/build_variant.sh all > /dev/null || echo "build of all failed"
git checkout -- hdl_graph_slam_amd/csrc
ls -la ab_libs/
