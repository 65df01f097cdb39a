#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "4 512" "2 512" "1 512" "4 768"; do
  set -- $cfg
  echo "== bench NDT lanes=$1 resident=$2"
  HGS_BATCH_LANES=$1 HGS_NDT_RESIDENT=$2 timeout 300 python bench.py --method NDT_OMP --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/v6_bench_l$1_r$2.log 2>&1; echo "exit $?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/v6_bench_l$1_r$2.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"], d["roofline"]["stage_ms_per_step"])
PY
done
echo "== odometry NDT"
timeout 300 python scripts/odometry_stream.py --method NDT_OMP --sensor HDL-64E --speed 3 --scans 16 --oracle-scans 0 > gpurun_out/v6_odo.log 2>&1; echo "exit $?"; tail -1 gpurun_out/v6_odo.log | cut -c1-330
