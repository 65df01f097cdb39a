#!/bin/bash
# Round 4 same-box A/B of library variants (ab_libs/<name>.so) over the four workloads the round's changes touch:
#   the metric batch (FAST_GICP, stage times), the NDT batch, config 2 (one cold HDL-32E align), config 3 at 3 m/s (NDT per-iteration latency).
#   gpurun -- 'VARIANTS="vH2 ndt_rep8 lm_lds greedy" bash scripts/r04_ab.sh'
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
for rep in $(seq 1 ${REPS:-1}); do for v in ${VARIANTS:-vH2}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  for W in ${WORKLOADS:-gicp ndt cfg2 cfg3}; do
    case $W in
      gicp) ARGS="--method FAST_GICP --steps 12 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1";;
      plane) ARGS="--method FAST_GICP --regularization PLANE --steps 12 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1";;
      ndt) ARGS="--method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1";;
      cfg2) ARGS="--config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1";;
      cfg3) ARGS="--config 3 --speed 3 --steps 40 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0";;
    esac
    echo -n "$v $W: "
    timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']; w = '$W'
    if w in ('gicp', 'plane', 'ndt'):
        print(r['value'], r['ms_per_step'], 'its', r['mean_iterations'], 'conv', r['converged'], 'best', r['best_candidate'], 'rmse', r['pose_rmse_vs_ground_truth']['translation_m'], 'cov', s['covariance'], 'lin', s['linearize'], 'err', s['error'], 'solve', s['solve'], 'fit', s['fitness'], 'idx', s['index'], 'launch us', r['roofline']['avg_launch_us'])
    elif w == 'cfg2':
        print(r['value'], r['ms_per_step'], 'warm p50', r['warm_align_ms']['p50'], 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
    else:
        print(r['value'], 'p50 ms', r['latency_ms']['p50'], 'p90', r['latency_ms']['p90'], 'its', r['mean_iterations'], 'us/it', r.get('us_per_iteration_p50'), 'track', r['trajectory_error_vs_ground_truth'])
"
  done
done; done 2>&1 | tee -a gpurun_out/r04_ab.log
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
