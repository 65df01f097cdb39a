#!/bin/bash
# quick A/B visit: GICP parity tests + the bench line (stage table) for the methods named in $METHODS
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "${KEXPR:-gicp or linearize or fitness}" 2>&1 | tail -3
for M in ${METHODS:-FAST_GICP}; do
  timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --seeds 1 --method $M 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('$M', r['value'], r['ms_per_step'], r.get('step_ms'), r['roofline']['avg_launch_us'], r['roofline']['stage_ms_per_step'])
"
done
