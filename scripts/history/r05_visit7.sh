#!/bin/bash
# Round 5, visit 7 (same as 6, with the pack pool): the host-packed point upload on config 3 (per-call split of a sweep) and config 2; GPU tests that touch the upload / prefilter / nn paths.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 300 python bench.py --config 3 --steps 40 --warmup 3 --no-cpu-baseline --oracle-sweeps 0 --seeds 1 > gpurun_out/r05_v7_config3.json 2> gpurun_out/r05_v7_config3.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_v7_config3.json"))
print("config3 p50", r["latency_ms"]["p50"], "value", r["value"], "its", r["mean_iterations"], "3mps", r["at_3_mps"]["latency_ms"]["p50"], r["at_3_mps"]["value"])
for k in ("kitti_prefilter_ndt_omp", "kitti_launch_fast_gicp"):
    print(k, r[k]["value"], r[k]["latency_ms"]["p50"], r[k]["trajectory_error_vs_ground_truth"])
for k, v in r["adapter_path"].items():
    if isinstance(v, dict):
        print(k, {m: (v[m]["p50_ms"], v[m]["max_ms"]) for m in ("c_abi", "adapter", "adapter_without_aligned_cloud", "pcl_align_alone")}, "ratio", v["adapter_over_c_abi_p50"], "calls", v["c_abi_calls"])
PY
timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print('config2', r['value'], r['step_ms']['p50'], 'warm', r['warm_align_ms']['p50'])"
timeout 900 python -m pytest tests/test_prefilter.py tests/test_odometry.py tests/test_hip_parity.py tests/test_adapter_cpp.py tests/test_integration_patch.py tests/test_map_cloud.py -m gpu -x -q 2>&1 | tail -5
