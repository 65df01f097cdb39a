#!/bin/bash
# Round 5, visit 8: config 3 after the cloud blocks became reusable across sweeps of slightly different sizes (and with the packed upload + pack pool).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 300 python bench.py --config 3 --steps 40 --warmup 3 --no-cpu-baseline --oracle-sweeps 0 --seeds 1 > gpurun_out/r05_v8_config3.json 2> gpurun_out/r05_v8_config3.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_v8_config3.json"))
print("config3 p50", r["latency_ms"], "value", r["value"], "its", r["mean_iterations"], "3mps", r["at_3_mps"]["latency_ms"]["p50"], r["at_3_mps"]["value"])
for k in ("kitti_prefilter_ndt_omp", "kitti_launch_fast_gicp"):
    print(k, r[k]["value"], r[k]["latency_ms"]["p50"], r[k]["trajectory_error_vs_ground_truth"])
for k, v in r["adapter_path"].items():
    if isinstance(v, dict):
        print(k, {m: (v[m]["p50_ms"], v[m]["max_ms"]) for m in ("c_abi", "adapter", "adapter_without_aligned_cloud", "adapter_with_eager_cpu_kdtree", "pcl_align_alone")}, "ratio", v["adapter_over_c_abi_p50"], "calls", v["c_abi_calls"])
PY
