#!/bin/bash
# Round 5 same-box A/B of library variants (ab_libs/<name>.so): the metric batch (FAST_GICP FROBENIUS / PLANE) in its timed configuration (lanes),
# with a rocprofv3 kernel trace of the lanes configuration per variant (k_gicp_solve / k_gicp_decide averages: the priority-inversion figure).
#   gpurun -- 'VARIANTS="base cw6 cw6p" bash scripts/r05_ab.sh'
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
export TMPDIR=/tmp
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
for rep in $(seq 1 ${REPS:-2}); do for v in ${VARIANTS:-base}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  for W in ${WORKLOADS:-gicp plane}; do
    case $W in
      gicp) ARGS="--method FAST_GICP --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1";;
      plane) ARGS="--method FAST_GICP --regularization PLANE --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --seeds 1";;
      ndt) ARGS="--method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1";;
      cfg2) ARGS="--config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1";;
      cfg3) ARGS="--config 3 --speed 3 --steps 40 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0 --no-kitti-records --no-adapter-record";;
    esac
    echo -n "$v $W: "
    timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']; w = '$W'
    if w in ('gicp', 'plane', 'ndt'):
        print(r['value'], r['ms_per_step'], 'p50', r['step_ms']['p50'], 'its', r['mean_iterations'], 'conv', r['converged'], 'best', r['best_candidate'], 'resident', r['resident_keyframes_value'], 'cov', s['covariance'], 'lin', s['linearize'], 'err', s['error'], 'solve', s['solve'], 'fit', s['fitness'], 'idx', s['index'])
    elif w == 'cfg2':
        print(r['value'], r['ms_per_step'], 'warm p50', r['warm_align_ms']['p50'], 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
    else:
        print(r['value'], 'p50 ms', r['latency_ms']['p50'], 'p90', r['latency_ms']['p90'], 'its', r['mean_iterations'], 'us/it', r.get('us_per_iteration_p50'))
"
  done
done; done 2>&1 | tee -a gpurun_out/r05_ab.log
if [ -n "${TRACE:-}" ]; then for v in ${VARIANTS:-base}; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  rm -rf /tmp/prof_$v
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o t -- python "$ROOT/bench.py" --method FAST_GICP --steps 3 --warmup 1 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1 > /tmp/prof_$v.json 2>/dev/null )
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v lanes kernel stats ($f)" | tee -a gpurun_out/r05_ab.log
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a gpurun_out/r05_ab.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r["Name"][:40], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2), "max us", round(float(r["MaxNs"]) / 1e3, 1), "total ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
done; fi
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
