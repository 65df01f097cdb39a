#!/bin/bash
# k_ndt_pass launch knobs re-measured on the round-4 kernel (the flush now adds up 44 KB of LDS lane slots: larger grabs may pay)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
run() {
  echo -n "$* : "
  env "$@" timeout 200 python bench.py --method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], 'its', r['mean_iterations'], 'best', r['best_candidate'], 'pass us', r['roofline']['avg_launch_us'])
"
}
{
if [ -n "${SWEEP2:-}" ]; then
for r in 256 384 448 512 576; do for l in 2 3; do run HGS_NDT_RESIDENT=$r HGS_BATCH_LANES=$l; done; done
run HGS_NDT_RESIDENT=512 HGS_NDT_CHUNK=12
run HGS_NDT_RESIDENT=512 HGS_NDT_CHUNK=6
else
run HGS_NDT_CHUNK=8
for c in 4 12 16 24; do run HGS_NDT_CHUNK=$c; done
for r in 512 640 1024; do run HGS_NDT_RESIDENT=$r; done
for l in 2 4; do run HGS_BATCH_LANES=$l; done
run HGS_NDT_CHUNK=16 HGS_BATCH_LANES=4
run HGS_NDT_CHUNK=8
fi
} 2>&1 | tee -a gpurun_out/r04_ndt_knobs.log
