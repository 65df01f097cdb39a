#!/bin/bash
# Kernel timeline of one sweep of the KITTI launch file's pipeline (device prefilter -> FAST_GICP) and of one config-2 registration, current library.
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
TAG=${TAG:-r06}
python scripts/probes/kitti_pipeline_probe.py 2>&1 | tail -1 | tee gpurun_out/${TAG}_kitti_probe.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/trace_kitti" -o t -- python "$ROOT/scripts/probes/kitti_pipeline_probe.py" > "$ROOT/gpurun_out/trace_kitti.log" 2>&1); echo "trace kitti exit $?"
f=$(find gpurun_out/trace_kitti -name "*kernel_trace.csv" | head -1)
g=$(find gpurun_out/trace_kitti -name "*kernel_stats.csv" | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python scripts/probes/kitti_pipeline_probe.py   (gaps include rocprofv3's own ~8 us per dispatch)"; echo; echo '```'; cat gpurun_out/${TAG}_kitti_probe.log; echo '```'; echo;
  [ -n "$g" ] && python scripts/prof_summary.py "$g"; echo; [ -n "$f" ] && python scripts/trace_timeline.py "$f" --max-rows 120; } > gpurun_out/${TAG}_kitti_timeline.md
find gpurun_out/trace_kitti -name "*.csv" -delete
if [ -z "${SKIP_CFG2:-}" ]; then
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/trace_cfg2" -o t -- python "$ROOT/bench.py" --config 2 --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 > "$ROOT/gpurun_out/trace_cfg2.log" 2>&1); echo "trace cfg 2 exit $?"
f=$(find gpurun_out/trace_cfg2 -name "*kernel_trace.csv" | head -1)
g=$(find gpurun_out/trace_cfg2 -name "*kernel_stats.csv" | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --seeds 1"; echo; [ -n "$g" ] && python scripts/prof_summary.py "$g"; echo; [ -n "$f" ] && python scripts/trace_timeline.py "$f" --max-rows 80; } > gpurun_out/${TAG}_config2_timeline.md
find gpurun_out/trace_cfg2 -name "*.csv" -delete
fi
tail -30 gpurun_out/${TAG}_kitti_timeline.md
