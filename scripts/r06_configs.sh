#!/bin/bash
# Round 6: the other BASELINE configurations as their own bench lines (gpurun_out/r06_bench_config{2..5}.json), config 4 on the PLANE line and on NDT_OMP,
# and the two launch shapes a 1-GPU box can exercise beyond the driver's: the process-group path at world size 1 (RCCL + the untimed trial step of the
# library's exchange) and the single-process mode with two engines on one device.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
for cfg in ${CONFIGS:-2 3 4 5}; do
  extra=""
  [ $cfg = 4 ] && extra="--fitness-max-range-variant"
  timeout 900 python bench.py --config $cfg --seeds ${SEEDS:-1} --cpu-single-thread $extra 2>gpurun_out/r06_bench_config$cfg.err | grep '^{' | tail -1 > gpurun_out/r06_bench_config$cfg.json
  echo "config $cfg exit $? $(cut -c1-300 gpurun_out/r06_bench_config$cfg.json)"
done
timeout 900 python bench.py --config 4 --regularization PLANE --seeds 1 2>gpurun_out/r06_bench_config4_plane.err | grep '^{' | tail -1 > gpurun_out/r06_bench_config4_plane.json
echo "config 4 PLANE $(cut -c1-200 gpurun_out/r06_bench_config4_plane.json)"
timeout 900 python bench.py --config 4 --method NDT_OMP --seeds 1 2>gpurun_out/r06_bench_config4_ndt.err | grep '^{' | tail -1 > gpurun_out/r06_bench_config4_ndt.json
echo "config 4 NDT $(cut -c1-200 gpurun_out/r06_bench_config4_ndt.json)"
HGS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1 2>gpurun_out/r06_bench_world1_rccl.err | grep '^{' | tail -1 > gpurun_out/r06_bench_world1_rccl.json
python -c "
import json; r = json.load(open('gpurun_out/r06_bench_world1_rccl.json')); print('world-1 process-group path:', r['value'], r['config']['exchange'], r['per_rank_ms_per_step'])"
HGS_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --single-process --candidates 32 --steps 10 --warmup 2 2>gpurun_out/r06_bench_single_process.err | grep '^{' | tail -1 > gpurun_out/r06_bench_single_process.json
python -c "
import json; r = json.load(open('gpurun_out/r06_bench_single_process.json')); print('single process, 2 engines on one device:', r['value'], r['ms_per_step'], r['config']['parallelism'])"
