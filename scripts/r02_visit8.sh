#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== comm + cpp + lifecycle tests"
timeout 900 python -m pytest tests/test_distributed.py tests/test_adapter_cpp.py tests/test_golden.py tests/test_hip_parity.py -m gpu -q --timeout 500 -p no:cacheprovider -k "sharded or cpp_loop or long_ndt or outlive or rccl" --durations=6 > gpurun_out/v8_pytest.log 2>&1; echo "exit $?"; tail -16 gpurun_out/v8_pytest.log
echo "== bench GICP (system fences gone from the progress mirror)"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/v8_bench_gicp.log 2>&1; echo "exit $?"
python - <<PY
import json
l=[x for x in open("gpurun_out/v8_bench_gicp.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"], d["roofline"]["stage_ms_per_step"])
PY
