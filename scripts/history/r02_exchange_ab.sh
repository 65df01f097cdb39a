#!/bin/bash
# the bench's sharded step at world size 1 over RCCL, through the library's own exchange and through torch.distributed
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for X in hgs torch; do
  HGS_BENCH_EXCHANGE=$X HGS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --seeds 1 > gpurun_out/exchange_$X.log 2>&1
  echo "exit $?"
  grep '^{' gpurun_out/exchange_$X.log | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print(r['value'], r['ms_per_step'], r['config']['exchange'], r['best_candidate'], r['converged'])
"
  grep -i "error\|Traceback\|failed" gpurun_out/exchange_$X.log | head -5
done
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "rccl or process_group or sharded" 2>&1 | tail -3
