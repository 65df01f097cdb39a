#!/bin/bash
# Round 6, fourteenth A/B: a finished single registration hands its result over in host-mapped memory (engine option early_result; no result kernel, no
# copy, no synchronisation, the rounds queued ahead drain behind the caller's back).  Bitwise probe, parity / odometry / adapter tests, kitti pipeline + config 2.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab14.log
echo "== bitwise: two-launch rounds (early result) vs four-launch rounds" | tee -a $LOG
timeout 600 python scripts/probes/fused_rounds_bits.py 40 2>&1 | tail -3 | tee -a $LOG
echo -n "tests: " | tee -a $LOG
timeout 2000 python -m pytest tests/test_hip_parity.py tests/test_odometry.py tests/test_loop_detector.py tests/test_adapter_cpp.py tests/test_integration_patch.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
for rep in 1 2 3; do for v in 0 1; do
  export HGS_ENGINE_OPTIONS="early_result=$v"
  echo -n "early_result=$v kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "early_result=$v cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
unset HGS_ENGINE_OPTIONS
