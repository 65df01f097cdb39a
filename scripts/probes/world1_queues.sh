#!/bin/bash
# the process-group path at world size 1 (torch's NCCL group + the library's RCCL communicator + the engine's lanes in one process) against the number of hardware queues
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1"
for q in 8 12 16 24; do for ex in hgs torch; do
  echo -n "GPU_MAX_HW_QUEUES=$q exchange=$ex: "
  GPU_MAX_HW_QUEUES=$q HGS_BENCH_EXCHANGE=$ex HGS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $A 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], r['config']['exchange'][:30])"
done; done 2>&1 | tee gpurun_out/r05_world1_queues.log
