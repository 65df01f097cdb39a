#!/bin/bash
# Round 5, visit 5: config 3 (8 m/s) with the adapter_path record and the per-call split of a sweep, config 2, on the final host path; quick metric check.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 300 python bench.py --config 3 --steps 40 --warmup 3 --no-kitti-records --no-cpu-baseline --oracle-sweeps 0 --seeds 1 > gpurun_out/r05_v5_config3.json 2> gpurun_out/r05_v5_config3.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_v5_config3.json"))
print("config3 p50", r["latency_ms"]["p50"], "value", r["value"], "its", r["mean_iterations"])
for k, v in r["adapter_path"].items():
    if isinstance(v, dict):
        print(k, {m: (v[m]["p50_ms"], v[m]["max_ms"]) for m in ("c_abi", "adapter", "adapter_without_aligned_cloud", "adapter_with_eager_cpu_kdtree", "pcl_align_alone")},
              "ratio", v["adapter_over_c_abi_p50"], "calls", v["c_abi_calls"], "pts", v["points_per_sweep"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --seeds 1 --no-ndt-record > gpurun_out/r05_v5_metric.json 2> gpurun_out/r05_v5_metric.err
python -c "
import json; r = json.load(open('gpurun_out/r05_v5_metric.json')); print('metric', r['value'], r['ms_per_step'], 'plane', r['fast_gicp_plane']['value'])"
