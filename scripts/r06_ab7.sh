#!/bin/bash
# Round 6, seventh A/B: k_knn_cov's pre-fill window for short packets (library base2 = before) x engine option knn_qpw_tiny (16- / 8-query packets for tiny launches)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab7.log
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
echo -n "new library, covariance / parity / odometry tests: " | tee -a $LOG
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_odometry.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -3 | tee -a $LOG
for rep in 1 2 3; do for v in base2:0 new:0 new:16 new:8; do
  lib=${v%%:*}; tiny=${v##*:}
  if [ "$lib" = base2 ]; then cp ab_libs/base2.so hdl_graph_slam_amd/lib/libhgs_hip.so; else cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so; fi
  export HGS_ENGINE_OPTIONS="knn_qpw_tiny=$tiny"
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
cp ab_libs/knnprobe.so hdl_graph_slam_amd/lib/libhgs_hip.so
timeout 600 python scripts/probes/knn_probe.py 2>&1 | tee gpurun_out/r06_knn_probe_after.log | grep -v "slow wave" | tee -a $LOG
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
