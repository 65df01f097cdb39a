#!/bin/bash
# Round 5, visit 10: k_bbox_count with one set of atomics per block; the process-group path with bench.py's new default of 16 hardware queues; upload tests.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 300 python bench.py --config 3 --steps 40 --warmup 3 --no-cpu-baseline --oracle-sweeps 0 --seeds 1 --no-kitti-records > gpurun_out/r05_v10_config3.json 2>/dev/null
python - <<'PY'
import json
r = json.load(open("gpurun_out/r05_v10_config3.json"))
print("config3", r["latency_ms"], "value", r["value"], "3mps", r["at_3_mps"]["latency_ms"]["p50"], r["at_3_mps"]["value"])
for k, v in r["adapter_path"].items():
    if isinstance(v, dict):
        print(k, {m: (v[m]["p50_ms"], v[m]["max_ms"]) for m in ("c_abi", "adapter", "adapter_without_aligned_cloud", "pcl_align_alone")}, "ratio", v["adapter_over_c_abi_p50"], "calls", v["c_abi_calls"])
PY
A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1"
HGS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $A 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r05_bench_world1_rccl.json
python -c "
import json; r = json.load(open('gpurun_out/r05_bench_world1_rccl.json')); print('world-1 process-group path:', r['value'], r['ms_per_step'], r['config']['exchange'], r['per_rank_ms_per_step'])"
timeout 600 python -m pytest tests/test_prefilter.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python "$GRAFT_REPO_ROOT/bench.py" --config 3 --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0 --no-kitti-records --no-adapter-record > /dev/null 2>&1; g=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); python "$GRAFT_REPO_ROOT/scripts/prof_summary.py" "$g" | grep "bbox\|pack_aos"
