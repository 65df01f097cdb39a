#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== quick parity"; timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "ndt_align or ndt_derivatives" > gpurun_out/v5_pytest.log 2>&1; echo "exit $?"; tail -2 gpurun_out/v5_pytest.log
for cfg in "4 512 0" "1 512 0" "4 512 4" "4 1024 0" "2 512 0"; do
  set -- $cfg
  echo "== bench NDT lanes=$1 resident=$2 chunk=$3"
  HGS_BATCH_LANES=$1 HGS_NDT_RESIDENT=$2 HGS_NDT_CHUNK=$3 timeout 300 python bench.py --method NDT_OMP --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/v5_bench_l$1_r$2_c$3.log 2>&1; echo "exit $?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/v5_bench_l$1_r$2_c$3.log") if x.startswith("{")]
d=json.loads(l[-1]); print("value", d["value"], "ms/step", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"], d["roofline"]["stage_ms_per_step"])
PY
done
for res in 512 256; do
  echo "== odometry NDT resident=$res"
  HGS_NDT_RESIDENT=$res timeout 300 python scripts/odometry_stream.py --method NDT_OMP --sensor HDL-64E --speed 3 --scans 16 --oracle-scans 0 > gpurun_out/v5_odo_r$res.log 2>&1; echo "exit $?"; tail -1 gpurun_out/v5_odo_r$res.log | cut -c1-330
done
echo "== rocprofv3 NDT bench (lanes 1)"
(cd /tmp && HGS_BATCH_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/v5_prof_ndt" -o bench -- python "$OLDPWD/bench.py" --method NDT_OMP --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/v5_prof_ndt.log" 2>&1); echo "prof exit $?"
f=$(find gpurun_out/v5_prof_ndt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -3 "$f" | cut -c1-60,330-420
