#!/bin/bash
# Round 6, fifth A/B (library variants): k_knn_cov with per-lane batched insertion rounds per leaf (-DHGS_KNN_BATCH=1), at 5 and at 4 waves per SIMD
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab5.log
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
for v in ${VARIANTS:-knnbatch knnbatch4}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  echo -n "$v parity: " | tee -a $LOG
  timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "covarianc or regulari or gicp_align" 2>&1 | tail -1 | tee -a $LOG
done
for rep in 1 2; do for v in base ${VARIANTS:-knnbatch knnbatch4}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  for W in gicp plane cfg2 cfg5; do
    case $W in
      gicp) ARGS="--method FAST_GICP --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1";;
      plane) ARGS="--method FAST_GICP --regularization PLANE --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --seeds 1";;
      cfg2) ARGS="--config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1";;
      cfg5) ARGS="--config 5 --steps 20 --warmup 3 --no-cpu-baseline --seeds 1";;
    esac
    echo -n "$v $W: " | tee -a $LOG
    timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'its', r.get('mean_iterations', r.get('iterations')), 'best', r.get('best_candidate'), {k: v for k, v in s.items() if v})
" | tee -a $LOG
  done
done; done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
