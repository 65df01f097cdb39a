#!/bin/bash
# Round 5, first GPU visit: the integration artefacts on the device (adapter with the lazy CPU tree, the patched reference factory / loop detector),
# config 3 with the new adapter_path record, and the metric line of the library that no longer sets GPU_MAX_HW_QUEUES itself.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 600 python -m pytest tests/test_adapter_cpp.py tests/test_integration_patch.py tests/test_loop_detector.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r05_v1_tests.log
timeout 300 python bench.py --config 3 --steps 40 --warmup 3 --no-kitti-records --no-cpu-baseline --oracle-sweeps 0 --seeds 1 > gpurun_out/r05_v1_config3.json 2> gpurun_out/r05_v1_config3.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_v1_config3.json"))
print("config3 p50", r["latency_ms"]["p50"], "value", r["value"])
for k, v in r["adapter_path"].items():
    if isinstance(v, dict):
        print(k, {m: (v[m]["p50_ms"], v[m]["max_ms"], v[m]["cpu_kdtree_builds"]) for m in ("c_abi", "adapter", "adapter_without_aligned_cloud", "adapter_with_eager_cpu_kdtree", "pcl_align_alone")},
              "ratio", v["adapter_over_c_abi_p50"], v["adapter_minus_pcl_align_over_c_abi_p50"], "pts", v["points_per_sweep"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --seeds 1 --no-plane-record --no-ndt-record > gpurun_out/r05_v1_metric.json 2> gpurun_out/r05_v1_metric.err
python -c "
import json; r = json.load(open('gpurun_out/r05_v1_metric.json')); print('metric', r['value'], r['ms_per_step'], r['step_ms'], 'resident', r['resident_keyframes_value'], r['roofline']['stage_ms_per_step'])"
