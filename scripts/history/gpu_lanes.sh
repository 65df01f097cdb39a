#!/bin/bash
# A/B of the number of batch lanes (HGS_BATCH_LANES) on the loop-closure bench; logs in gpurun_out/lanes.log
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
: > gpurun_out/lanes.log
for method in ${METHODS:-FAST_GICP NDT_OMP}; do
  for lanes in ${LANES:-1 2 4}; do
    echo "== $method lanes=$lanes" >> gpurun_out/lanes.log
    HGS_BATCH_LANES=$lanes timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --method $method 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], r['resident_keyframes_value'], r['roofline']['stage_ms_per_step'])
" >> gpurun_out/lanes.log
  done
done
cat gpurun_out/lanes.log
