#!/bin/bash
# Same-box A/B of library variants: build each variant, copy it to ab_libs/<name>.so (untracked, travels with gpurun), then
#   gpurun -- "VARIANTS=\"v0 v1\" METHODS=FAST_GICP bash scripts/r02_ab_libs.sh"   (box-to-box variation is +-2 %: compare within one visit)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for v in ${VARIANTS:-v0 v1}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  for M in ${METHODS:-FAST_GICP}; do
  echo -n "$v $M: "
  timeout 300 python bench.py --method $M --steps 10 --warmup 2 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); s = r['roofline']['stage_ms_per_step']; print(r['value'], r['ms_per_step'], 'cov', s['covariance'], 'lin', s['linearize'], 'fit', s['fitness'])
"
  done
done; done
