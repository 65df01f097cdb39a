#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for cap in 8 2 16 8; do
  echo "== bench NDT (64 cand) GSS cap=$cap"
  HGS_NDT_CHUNK=$cap timeout 300 python bench.py --method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --seeds 1 > gpurun_out/v10_bench_c$cap.log 2>&1; echo "exit $?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/v10_bench_c$cap.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"])
PY
done
for cap in 8 1 8; do
echo "== config 3 cap=$cap"
HGS_NDT_CHUNK=$cap timeout 300 python bench.py --config 3 --no-cpu-baseline --seeds 1 > gpurun_out/v10_cfg3_c$cap.log 2>&1; echo "exit $?"
python - <<PY
import json
l=[x for x in open("gpurun_out/v10_cfg3_c$cap.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], d["latency_ms"], d["mean_iterations"])
PY
done
