#!/bin/bash
# Round 6, eighth A/B: k_knn_cov short packets behind a full pre-fill window (template WINDOW) — engine options knn_qpw_tiny : knn_tiny_below; library base2 = before
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab8.log
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
echo -n "new library, covariance / parity / odometry tests: " | tee -a $LOG
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_odometry.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
echo -n "knn_qpw_tiny=8,knn_tiny_below=200000, parity / odometry tests: " | tee -a $LOG
HGS_ENGINE_OPTIONS="knn_qpw_tiny=8,knn_tiny_below=200000" timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_odometry.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
for rep in 1 2 3; do for v in ${COMBOS:-base2:0:0 new:0:0 new:8:32768 new:16:32768 new:8:200000 new:16:200000 new:24:200000}; do
  lib=${v%%:*}; rest=${v#*:}; tiny=${rest%%:*}; below=${rest##*:}
  if [ "$lib" = base2 ]; then cp ab_libs/base2.so hdl_graph_slam_amd/lib/libhgs_hip.so; export HGS_ENGINE_OPTIONS="knn_qpw_tiny=0"
  else cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so; export HGS_ENGINE_OPTIONS="knn_qpw_tiny=$tiny,knn_tiny_below=$below"; fi
  echo -n "$v kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "$v cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
unset HGS_ENGINE_OPTIONS
for v in base2 new; do
  if [ "$v" = base2 ]; then cp ab_libs/base2.so hdl_graph_slam_amd/lib/libhgs_hip.so; else cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so; fi
  for W in gicp plane; do
    case $W in
      gicp) ARGS="--method FAST_GICP --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1";;
      plane) ARGS="--method FAST_GICP --regularization PLANE --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --seeds 1";;
    esac
    echo -n "$v $W: " | tee -a $LOG
    timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'its', r.get('mean_iterations'), 'best', r.get('best_candidate'), {k: v for k, v in s.items() if v})
" | tee -a $LOG
  done
done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
