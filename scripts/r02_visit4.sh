#!/bin/bash
# Round 2, GPU visit 4: NDT pass v3 (work queue) — lanes x resident blocks, odometry latency, kernel trace, test durations.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "4 512" "1 512" "4 256" "4 1024" "2 512" "1 1024"; do
  set -- $cfg
  echo "== bench NDT lanes=$1 resident=$2"
  HGS_BATCH_LANES=$1 HGS_NDT_RESIDENT=$2 timeout 300 python bench.py --method NDT_OMP --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/v4_bench_l$1_r$2.log 2>&1; echo "exit $?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/v4_bench_l$1_r$2.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"], d["roofline"]["stage_ms_per_step"])
PY
done
for res in 512 256; do
  echo "== odometry NDT resident=$res"
  HGS_NDT_RESIDENT=$res timeout 300 python scripts/odometry_stream.py --method NDT_OMP --sensor HDL-64E --speed 3 --scans 16 --oracle-scans 0 > gpurun_out/v4_odo_r$res.log 2>&1; echo "exit $?"; tail -1 gpurun_out/v4_odo_r$res.log | cut -c1-330
done
echo "== rocprofv3 NDT bench (lanes 1)"
(cd /tmp && HGS_BATCH_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/v4_prof_ndt" -o bench -- python "$OLDPWD/bench.py" --method NDT_OMP --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/v4_prof_ndt.log" 2>&1); echo "prof exit $?"
f=$(find gpurun_out/v4_prof_ndt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -3 "$f" | cut -c1-200
echo "== NDT tests with durations"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size.py tests/test_odometry.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "ndt or NDT" --durations=12 > gpurun_out/v4_pytest_ndt.log 2>&1
echo "pytest exit $?"; tail -22 gpurun_out/v4_pytest_ndt.log
