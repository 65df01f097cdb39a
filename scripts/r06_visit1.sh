#!/bin/bash
# Round 6, first GPU visit: the reference-code pins through the C-ABI, the NDT plan ordering fix, the whole -m gpu suite, and this box's baseline lines.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 900 python -m pytest tests/test_reference_code_pins.py tests/test_loop_detector.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06_v1_new_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --seeds 1 > gpurun_out/r06_v1_metric.json 2> gpurun_out/r06_v1_metric.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06_v1_metric.json"))
print("metric", r["value"], r["ms_per_step"], r["roofline"]["stage_ms_per_step"])
for k in ("fast_gicp_plane", "ndt_omp"):
    if k in r: print(k, r[k].get("value"), r[k].get("ms_per_step"), r[k].get("roofline", {}).get("stage_ms_per_step"))
PY
timeout 2400 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r06_v1_pytest_gpu.log
