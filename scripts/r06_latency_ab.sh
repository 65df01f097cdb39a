#!/bin/bash
# Round 6, the odometry step: same-box A/B by environment knob of the KITTI launch file's pipeline (device prefilter -> FAST_GICP; per-call p50s of
# scripts/probes/kitti_pipeline_probe.py) and of config 2 (one cold HDL-32E registration).
#   gpurun -- 'COMBOS="0:0 1:0 1:16 1:8" bash scripts/r06_latency_ab.sh'      (HGS_RESIDENT_DESCS : HGS_KNN_QPW_TINY)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/${LOG:-r06_latency_ab}.log
if [ -n "${PRETEST:-}" ]; then timeout 1200 python -m pytest tests/test_prefilter.py tests/test_odometry.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -4 | tee -a $LOG; fi
for rep in $(seq 1 ${REPS:-2}); do for combo in ${COMBOS:-1:16}; do
  export HGS_ENGINE_OPTIONS="resident_descs=${combo%%:*},knn_qpw_tiny=${combo##*:}"
  echo -n "$HGS_ENGINE_OPTIONS kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "$HGS_ENGINE_OPTIONS cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
