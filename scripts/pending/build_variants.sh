#!/bin/bash
# CPU side of the A/B of scripts/pending: ab_libs/vH.so (the tree as it is), ab_libs/<patch>.so per patch, and the two combinations that
# apply together (the two reduction patches exclude each other): items32_scalar.so, items32_mfma.so.
# GPU side:  gpurun -- 'VARIANTS="vH ndt_pass_items32 ndt_reduction_scalar_scales ndt_reduction_mfma items32_mfma" bash scripts/r03_ab_ndt.sh'
set -eu
cd "$(dirname "$0")/../.."
git diff --quiet -- hdl_graph_slam_amd/csrc oracle tests/emul || { echo "uncommitted changes in the patched directories"; exit 1; }
restore() { git checkout -- hdl_graph_slam_amd/csrc oracle tests/emul; }
trap restore EXIT
scripts/build_variant.sh vH > /dev/null
for p in scripts/pending/*.patch; do
  n=$(basename "$p" .patch)
  git apply "$p"
  scripts/build_variant.sh "$n" > /dev/null || echo "build of $n failed"
  restore
done
for r in scalar_scales mfma; do
  git apply scripts/pending/ndt_pass_items32.patch scripts/pending/ndt_reduction_$r.patch
  scripts/build_variant.sh items32_$r > /dev/null || echo "build of items32_$r failed"
  restore
done
ls -la ab_libs/
