#!/bin/bash
# Round 5, visit 9: the process-group path at world size 1 (RCCL exchange through the C-ABI) after the comm_wait backoff cap went from 1 ms to 50 us, next to
# the torch exchange and the plain single-rank batch on the same box; timelines of one registration (configs 2 and 3) on the final library.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1"
run() { python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('$1', r['value'], r['ms_per_step'], r['config']['exchange'], r['per_rank_ms_per_step'])"; }
for rep in 1 2; do
  timeout 300 python bench.py $A 2>/dev/null | run "single rank, no process group:"
  HGS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $A 2>/dev/null | tee gpurun_out/r05_bench_world1_rccl.log | run "world 1, hgs exchange:"
  HGS_BENCH_EXCHANGE=torch HGS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $A 2>/dev/null | run "world 1, torch exchange:"
done 2>&1 | tee gpurun_out/r05_world1.log
grep '^{' gpurun_out/r05_bench_world1_rccl.log | tail -1 > gpurun_out/r05_bench_world1_rccl.json
bash scripts/r05_timelines.sh
