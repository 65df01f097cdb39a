#!/bin/bash
# RECORD OF A DROPPED EXPERIMENT: the HGS_STAGGER_COV code path this script measured is not in the tree (it lost: profiles/r04_ab4_staggered_cov.log).
# Staggered per-lane covariance launches of a cold FAST_GICP batch against the one launch in front of the LM loop (HGS_STAGGER_COV=0)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
run() {
  echo -n "$* : "
  env "$@" timeout 300 python bench.py $ARGS --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], 'its', r['mean_iterations'], 'conv', r['converged'], 'best', r['best_candidate'], 'rmse', r['pose_rmse_vs_ground_truth']['translation_m'], 'resident', r.get('resident_keyframes_value'), 'p10-p90', r['step_ms']['p10'], r['step_ms']['p90'])
"
}
{
for rep in 1 2; do
ARGS="--method FAST_GICP --steps 20 --warmup 3"; echo "== metric FAST_GICP"; run HGS_STAGGER_COV=0; run HGS_STAGGER_COV=1; run HGS_STAGGER_COV=1 HGS_BATCH_LANES=3; run HGS_STAGGER_COV=1 HGS_BATCH_LANES=2
ARGS="--method FAST_GICP --regularization PLANE --steps 20 --warmup 3"; echo "== PLANE"; run HGS_STAGGER_COV=0; run HGS_STAGGER_COV=1
done
ARGS="--config 4 --steps 8 --warmup 2"; echo "== config 4"; run HGS_STAGGER_COV=0; run HGS_STAGGER_COV=1
} 2>&1 | tee gpurun_out/r04_stagger.log
