#!/bin/bash
# Round 6, second A/B: (1) PLANE covariances as search + k_cov_regularize (HGS_COV_SPLIT=1) vs the eigen-decomposition inline (0); (2) the seed grid's
# finest cell (library variants ab_libs/{base,seed8,seed16}.so = 0.25 / 0.125 / 0.0625 m) on the NDT batch's getFitnessScore; (3) config 5 / config 2 PLANE.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/${LOG:-r06_ab2}.log
line() { python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'its', r.get('mean_iterations', r.get('iterations')), 'best', r.get('best_candidate'), {k: v for k, v in s.items() if v})
"; }
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "covarianc or regulari" 2>&1 | tail -3 | tee -a $LOG
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
for rep in 1 2; do
  for split in 0 1; do
    echo -n "cov_split=$split plane: " | tee -a $LOG
    HGS_ENGINE_OPTIONS=cov_split=$split timeout 300 python bench.py --method FAST_GICP --regularization PLANE --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | line | tee -a $LOG
    echo -n "cov_split=$split cfg5 plane: " | tee -a $LOG
    HGS_ENGINE_OPTIONS=cov_split=$split timeout 300 python bench.py --config 5 --regularization PLANE --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 2>/dev/null | line | tee -a $LOG
  done
  for v in base seed8 seed16; do
    cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
    echo -n "$v ndt: " | tee -a $LOG
    timeout 300 python bench.py --method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | line | tee -a $LOG
  done
  cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
done
