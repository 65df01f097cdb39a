#!/bin/bash
# Same-box A/B of library variants (ab_libs/<name>.so, scripts/build_variant.sh):
#   gpurun -- 'VARIANTS="v1 v4" METHODS="FAST_GICP NDT_OMP" bash scripts/r03_ab.sh'       (box-to-box variation is +-2 %: compare within one visit)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
for rep in $(seq 1 ${REPS:-2}); do for v in ${VARIANTS:-v0 v1}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  for M in ${METHODS:-FAST_GICP}; do
  echo -n "$v $M: "
  timeout 300 python bench.py --method $M --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1 ${BENCH_FLAGS:---mild-set} 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); s = r['roofline']['stage_ms_per_step']; print(r['value'], r['ms_per_step'], 'its', r['mean_iterations'], 'cov', s['covariance'], 'lin', s['linearize'], 'err', s['error'], 'solve', s['solve'], 'fit', s['fitness'], 'idx', s['index'])
"
  done
done; done 2>&1 | tee -a gpurun_out/r03_ab.log
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
