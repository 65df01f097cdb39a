#!/bin/bash
# Round 3, last visit: what the driver runs at round end — smoke, pytest -m gpu, the default bench command — on the final tree.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -p no:cacheprovider > gpurun_out/r03_final_pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_final_pytest_gpu.log | tail -6
echo "== bench (driver's command)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03_final_bench.err | grep '^{' > gpurun_out/r03_final_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r03_final_bench.json')); print('driver cmd:', r['value'], r['ms_per_step'], r['mean_iterations'], 'frac', r['roofline']['frac'], r['roofline']['kernel'], r['roofline']['avg_launch_us'], 'traffic', r['roofline']['traffic'], 'alg', r['roofline']['algorithmic_bytes_per_launch'], '| ndt', r['ndt_omp']['value'], r['ndt_omp']['roofline']['frac'], r['ndt_omp']['roofline']['traffic'], '| r02 set', r['r02_candidate_set']['value'], '| cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])"
