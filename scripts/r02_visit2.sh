#!/bin/bash
# Round 2, GPU visit 2: NDT pass v2 — how the integer totals reach HBM (copies per XCD, blocks per problem).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== quick parity"; timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "ndt_align or ndt_derivatives" > gpurun_out/v2_pytest.log 2>&1; echo "exit $?"; tail -3 gpurun_out/v2_pytest.log
for cfg in "8 512" "1 512" "8 1024" "8 256" "8 100000" "1 100000"; do
  set -- $cfg
  echo "== bench NDT copies=$1 resident=$2"
  HGS_NDT_COPIES=$1 HGS_NDT_RESIDENT=$2 timeout 300 python bench.py --method NDT_OMP --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/v2_bench_c$1_r$2.log 2>&1; echo "exit $?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/v2_bench_c$1_r$2.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"], d["roofline"]["stage_ms_per_step"])
PY
done
for blocks in 0 256 128 64; do
  echo "== odometry NDT blocks=$blocks"
  HGS_NDT_BLOCKS=$blocks timeout 300 python scripts/odometry_stream.py --method NDT_OMP --sensor HDL-64E --speed 3 --scans 16 --oracle-scans 0 > gpurun_out/v2_odo_b$blocks.log 2>&1; echo "exit $?"; tail -1 gpurun_out/v2_odo_b$blocks.log | cut -c1-400
done
echo "== rocprofv3 NDT bench (default knobs)"
(cd /tmp && HGS_BATCH_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/v2_prof_ndt" -o bench -- python "$OLDPWD/bench.py" --method NDT_OMP --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/v2_prof_ndt.log" 2>&1); echo "prof exit $?"
f=$(find gpurun_out/v2_prof_ndt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-200
