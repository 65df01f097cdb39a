#!/bin/bash
# A/B of the NDT pass knobs on the 64-candidate bench workload (HGS_NDT_RESIDENT blocks per launch, HGS_NDT_CHUNK items per grab)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
: > gpurun_out/ndt_knobs.log
for res in ${RESIDENT:-512 768 1024}; do for ch in ${CHUNK:-8 16}; do
  echo -n "resident=$res chunk=$ch  " >> gpurun_out/ndt_knobs.log
  HGS_NDT_RESIDENT=$res HGS_NDT_CHUNK=$ch timeout 300 python bench.py --method NDT_OMP --steps 6 --warmup 2 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], r['step_ms']['p50'])
" >> gpurun_out/ndt_knobs.log
done; done
cat gpurun_out/ndt_knobs.log
