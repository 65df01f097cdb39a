#!/bin/bash
# Builds a variant of libhgs_hip.so with extra -D flags into ab_libs/<name>.so (untracked; travels with gpurun) for same-box A/B runs:
#   scripts/build_variant.sh v1 -DHGS_LINEARIZE_WAVES=6
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p ab_libs/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for src in hgs_sort hgs_kernels hgs_engine hgs_comm; do
  if [ "$src" = hgs_sort ] && [ -f hdl_graph_slam_amd/lib/hgs_sort.o ]; then cp hdl_graph_slam_amd/lib/hgs_sort.o ab_libs/obj_$name/; continue; fi
  hipcc $FLAGS "$@" -c hdl_graph_slam_amd/csrc/$src.hip -o ab_libs/obj_$name/$src.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/$name.so ab_libs/obj_$name/*.o -ldl
rm -rf ab_libs/obj_$name
ls -la ab_libs/$name.so
