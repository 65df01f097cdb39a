#!/bin/bash
# Lean end-of-round visit: smoke, the bench line (with cpu_baseline), one rocprofv3 kernel trace of the bench command with
# whole-device launches (HGS_BATCH_LANES=1: the launch shape bench.py's HIP-event roofline measurement times).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" | tee -a gpurun_out/bench.log; tail -2 gpurun_out/bench.log
(cd /tmp && HGS_BATCH_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_lanes1" -o bench -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_lanes1.log" 2>&1); echo "prof exit $?"
f=$(find gpurun_out/prof_lanes1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
