#!/bin/bash
# Round 6 same-box A/B by environment knob (one library): k_knn_cov gather mode 2 (leaf-wise sums) / 3 (neighbour-wise sums) x seed grid off / on,
# on the metric batch (FAST_GICP FROBENIUS / PLANE / NDT_OMP) and config 2.
#   gpurun -- 'COMBOS="0:2 0:3 1:2 1:3" bash scripts/r06_ab.sh'
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
export TMPDIR=/tmp
if [ -n "${PRETEST:-}" ]; then timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "$PRETEST" 2>&1 | tail -5 | tee -a gpurun_out/${LOG:-r06_ab}.log; fi
for rep in $(seq 1 ${REPS:-2}); do for combo in ${COMBOS:-1:3}; do
  export HGS_ENGINE_OPTIONS="seed_grid=${combo%%:*},knn_replay=${combo##*:}"
  for W in ${WORKLOADS:-gicp plane ndt cfg2}; do
    case $W in
      gicp) ARGS="--method FAST_GICP --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1";;
      plane) ARGS="--method FAST_GICP --regularization PLANE --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --seeds 1";;
      ndt) ARGS="--method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1";;
      cfg2) ARGS="--config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1";;
      cfg5) ARGS="--config 5 --steps 20 --warmup 3 --no-cpu-baseline --seeds 1";;
      cfg3) ARGS="--config 3 --speed 3 --steps 40 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0 --no-kitti-records --no-adapter-record";;
    esac
    echo -n "$HGS_ENGINE_OPTIONS $W: "
    timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']; w = '$W'
    if w in ('gicp', 'plane', 'ndt'):
        print(r['value'], r['ms_per_step'], 'p50', r['step_ms']['p50'], 'its', r['mean_iterations'], 'conv', r['converged'], 'best', r['best_candidate'], 'cov', s['covariance'], 'lin', s['linearize'], 'err', s['error'], 'solve', s['solve'], 'fit', s['fitness'], 'idx', s['index'])
    elif w in ('cfg2', 'cfg5'):
        print(r['value'], r['ms_per_step'], 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
    else:
        print(r['value'], 'p50 ms', r['latency_ms']['p50'], 'p90', r['latency_ms']['p90'], 'its', r['mean_iterations'], 'us/it', r.get('us_per_iteration_p50'))
"
  done
done; done 2>&1 | tee -a gpurun_out/${LOG:-r06_ab}.log
