#!/bin/bash
# Round 5: how the lanes of a batch are released (HGS_LANE_START: 0 = all by one event at the same instant, 1 = host synchronises first and enqueues lane after
# lane as rounds 1-4 did, 2 = lane i + 1 behind lane i's first point kernel) — same library, same box, interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
for rep in 1 2 3; do for W in gicp plane ndt; do for m in 0 1 2; do
  [ $W = ndt ] && [ $m = 2 ] && continue
  case $W in
    gicp) ARGS="--method FAST_GICP --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1";;
    plane) ARGS="--method FAST_GICP --regularization PLANE --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --seeds 1";;
    ndt) ARGS="--method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1";;
  esac
  echo -n "lane_start=$m $W: "
  HGS_LANE_START=$m timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], 'p50', r['step_ms']['p50'], 'its', r['mean_iterations'], 'conv', r['converged'], 'best', r['best_candidate'])"
done; done; done 2>&1 | tee gpurun_out/r05_lane_start.log
