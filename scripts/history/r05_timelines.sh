#!/bin/bash
# Kernel timelines of ONE registration (config 2: cold FAST_GICP pair; config 3 at 3 m/s: NDT_OMP odometry sweep) on the current library
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
for cfg in 2 3; do
  extra=""; [ $cfg = 3 ] && extra="--no-kitti-records --no-adapter-record"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/trace_cfg$cfg" -o t -- python "$ROOT/bench.py" --config $cfg $extra --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0 > "$ROOT/gpurun_out/trace_cfg$cfg.log" 2>&1); echo "trace cfg $cfg exit $?"
  f=$(find gpurun_out/trace_cfg$cfg -name "*kernel_trace.csv" | head -1)
  g=$(find gpurun_out/trace_cfg$cfg -name "*kernel_stats.csv" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg $extra --steps 20 --warmup 3 --no-cpu-baseline --seeds 1 --oracle-sweeps 0   (gaps include rocprofv3's own ~8 us per dispatch)"; echo; echo '```'; grep '^{' gpurun_out/trace_cfg$cfg.log | tail -1 | cut -c1-1200; echo '```'; echo;
    [ -n "$g" ] && python scripts/prof_summary.py "$g"; echo; [ -n "$f" ] && python scripts/trace_timeline.py "$f" --max-rows 60; } > gpurun_out/r05_config${cfg}_timeline.md
  find gpurun_out/trace_cfg$cfg -name "*.csv" -delete
done
grep "k_gicp_\|k_knn\|k_ndt_pass" gpurun_out/r05_config2_timeline.md gpurun_out/r05_config3_timeline.md | grep "^[^:]*:| hgs" | cut -c1-140
