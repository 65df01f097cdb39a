#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "4 2" "4 0" "4 1" "4 3" "4 2"; do
  set -- $cfg
  echo "== bench NDT lanes=$1 chunk=$2"
  HGS_BATCH_LANES=$1 HGS_NDT_CHUNK=$2 timeout 300 python bench.py --method NDT_OMP --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/v7_bench_l$1_c$2.log 2>&1; echo "exit $?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/v7_bench_l$1_c$2.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"])
PY
done
for ch in 1 0 1; do
echo "== odometry NDT chunk=$ch"
HGS_NDT_CHUNK=$ch timeout 300 python scripts/odometry_stream.py --method NDT_OMP --sensor HDL-64E --speed 3 --scans 16 --oracle-scans 0 > gpurun_out/v7_odo_c$ch.log 2>&1; echo "exit $?"; tail -1 gpurun_out/v7_odo_c$ch.log | cut -c80-200
done
