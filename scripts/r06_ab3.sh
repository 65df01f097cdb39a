#!/bin/bash
# Round 6, third A/B: fused LM tails (engine option fused_tails=0/1) on the KITTI launch file's pipeline and config 2; the seed grid with 0.5 m finest cells (ab_libs/seed2.so) on the NDT batch.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab3.log
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_odometry.py tests/test_loop_detector.py tests/test_integration_patch.py tests/test_adapter_cpp.py -m gpu -x -q 2>&1 | tail -4 | tee -a $LOG
for rep in 1 2 3; do for f in 0 1; do
  export HGS_ENGINE_OPTIONS=fused_tails=$f
  echo -n "fused=$f kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "fused=$f cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
unset HGS_ENGINE_OPTIONS
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
for rep in 1 2; do for v in base seed2; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  echo -n "$v ndt: " | tee -a $LOG
  timeout 300 python bench.py --method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
