#!/bin/bash
# How many levels of the Hilbert curve does the index sort have to compare?  (16 = all 48 bits, rounds 1-3)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
for rep in 1 2; do for L in 16 13 12 11 10 9; do
  echo -n "levels $L: "
  HGS_HILBERT_LEVELS=$L timeout 200 python bench.py --method FAST_GICP --steps 12 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); s = r['roofline']['stage_ms_per_step']; print(r['value'], r['ms_per_step'], 'its', r['mean_iterations'], 'best', r['best_candidate'], 'rmse', r['pose_rmse_vs_ground_truth']['translation_m'], 'cov', s['covariance'], 'lin', s['linearize'], 'fit', s['fitness'], 'idx', s['index'])
"
done; done 2>&1 | tee gpurun_out/r04_hilbert_levels.log
HGS_HILBERT_LEVELS=11 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
