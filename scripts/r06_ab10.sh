#!/bin/bash
# Round 6, tenth / eleventh A/B: the LM round of a single registration in two launches.  FUSED=fused_tails: the last block of a problem runs the control step
# in the point kernel's tail (fence-free ticket; profiles/r06_ab10_fused_tails_fence_free.{log,patch}); FUSED=fused_rounds (the library's): the control steps
# replicated in every block.  Bitwise check, parity tests, kitti pipeline + config 2.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
FUSED=${FUSED:-fused_rounds}
LOG=gpurun_out/${LOGNAME_AB:-r06_ab11}.log
echo "== bitwise: $FUSED=1 vs four-launch rounds" | tee -a $LOG
timeout 600 python scripts/probes/fused_rounds_bits.py 40 2>&1 | tail -12 | tee -a $LOG
echo -n "$FUSED=1, parity / odometry tests: " | tee -a $LOG
HGS_ENGINE_OPTIONS="$FUSED=1" timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_odometry.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
for rep in 1 2 3; do for v in ${COMBOS:-0 1}; do
  export HGS_ENGINE_OPTIONS="$FUSED=$v"
  echo -n "$FUSED=$v kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "$FUSED=$v cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
unset HGS_ENGINE_OPTIONS
