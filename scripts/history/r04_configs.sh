#!/bin/bash
# Round 4: the other BASELINE configurations as their own bench lines (gpurun_out/r04_bench_config{2..5}.json) and the driver's command once more
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
for cfg in ${CONFIGS:-2 3 4 5}; do
  extra=""
  [ $cfg = 4 ] && extra="--fitness-max-range-variant"
  timeout 900 python bench.py --config $cfg --seeds ${SEEDS:-1} --cpu-single-thread $extra 2>gpurun_out/r04_bench_config$cfg.err | grep '^{' | tail -1 > gpurun_out/r04_bench_config$cfg.json
  echo "config $cfg exit $? $(cut -c1-400 gpurun_out/r04_bench_config$cfg.json)"
done
