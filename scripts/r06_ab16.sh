#!/bin/bash
# Round 6, sixteenth A/B: how far the host runs ahead of a single registration that hands its result over early (engine option early_run_ahead: 2 / 3 / 4 rounds).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab16.log
for rep in 1 2 3; do for v in 2 3 4; do
  export HGS_ENGINE_OPTIONS="early_run_ahead=$v"
  echo -n "early_run_ahead=$v kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "early_run_ahead=$v cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln)
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'))
" | tee -a $LOG
done; done
unset HGS_ENGINE_OPTIONS
