#!/bin/bash
# Round 3, visit 3: gather probe, the whole GPU suite, the default bench line (new candidate set + NDT sub-record).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== gather probe"; timeout 120 scripts/probes/gather_probe 2>&1 | tee gpurun_out/r03_gather_probe.log
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest_gpu.log | tail -15
echo "== bench (default command)"
timeout 900 python bench.py --steps 20 > gpurun_out/r03_bench_default.log 2>&1; echo "bench exit $?"; tail -c 6000 gpurun_out/r03_bench_default.log
echo "== bench (mild set, for continuity with r02)"
timeout 600 python bench.py --steps 20 --mild-set --no-ndt-record --no-cpu-baseline --seeds 1 > gpurun_out/r03_bench_mild.log 2>&1; echo "bench exit $?"; tail -c 1500 gpurun_out/r03_bench_mild.log
