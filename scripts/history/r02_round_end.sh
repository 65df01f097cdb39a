#!/bin/bash
# Round 2: what the driver runs at round end (pytest -m gpu, smoke, bench) + durations
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/v11_smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/v11_smoke.log
echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider --durations=15 > gpurun_out/v11_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/v11_pytest_gpu.log
echo "== bench (driver command)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/v11_bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/v11_bench.log | cut -c1-600
