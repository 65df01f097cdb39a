#!/bin/bash
# Lanes x hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at initialisation; default 4) on the three lines
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
run() {
  echo -n "$M | $* : "
  env "$@" timeout 200 python bench.py $M --steps 16 --warmup 3 --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], 'best', r['best_candidate'])
"
}
{
for M in "--method FAST_GICP" "--method FAST_GICP --regularization PLANE" "--method NDT_OMP"; do
  run GPU_MAX_HW_QUEUES=4
  run GPU_MAX_HW_QUEUES=8
  for l in 5 6 8; do run GPU_MAX_HW_QUEUES=8 HGS_BATCH_LANES=$l; done
  run GPU_MAX_HW_QUEUES=8 HGS_BATCH_LANES=4
done
} 2>&1 | tee gpurun_out/r04_lanes_queues.log
