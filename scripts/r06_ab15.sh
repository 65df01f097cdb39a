#!/bin/bash
# Round 6, fifteenth A/B: rocPRIM's block-and-merge sort with 4096-item blocks for sorts of at most 256 k keys (hgs_sort.hip; ab_libs/sort_default.so = the
# library's own configuration, 1024-item blocks).  Prefilter / odometry / index tests on the new configuration, kitti pipeline + config 2.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab15.log
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
echo -n "tests (4096-item blocks): " | tee -a $LOG
timeout 1500 python -m pytest tests/test_prefilter.py tests/test_hip_parity.py tests/test_odometry.py tests/test_map_cloud.py -m gpu -x -q 2>&1 | tail -12 | grep -E "passed|failed|error" | tee -a $LOG
for rep in 1 2 3; do for v in sort_default big_blocks; do
  if [ "$v" = sort_default ]; then cp ab_libs/sort_default.so hdl_graph_slam_amd/lib/libhgs_hip.so; else cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so; fi
  echo -n "$v kitti: " | tee -a $LOG
  timeout 300 python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee -a $LOG
  echo -n "$v cfg2: " | tee -a $LOG
  timeout 300 python bench.py --config 2 --steps 400 --warmup 20 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'p50', r.get('step_ms', {}).get('p50'), 'warm p50', r.get('warm_align_ms', {}).get('p50'), 'its', r.get('iterations'), 'stages', {k: v for k, v in s.items() if v})
" | tee -a $LOG
done; done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
