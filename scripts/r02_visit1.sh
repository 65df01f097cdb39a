#!/bin/bash
# Round 2, GPU visit 1: the rebuilt NDT pass on hardware — parity to convergence, batch throughput (source order A/B), kernel
# stats.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== ndt parity (tests)"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_full_size.py tests/test_odometry.py -m gpu -q -x --timeout 900 --timeout-method=thread -p no:cacheprovider -k "ndt or NDT" > gpurun_out/v1_pytest_ndt.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/v1_pytest_ndt.log; tail -25 gpurun_out/v1_pytest_ndt.log
for sort in -1 1; do
  echo "== bench NDT_OMP sort=$sort"
  HGS_NDT_SORT=$sort timeout 600 python bench.py --method NDT_OMP --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/v1_bench_ndt_sort$sort.log 2>&1; echo "exit $?"; tail -2 gpurun_out/v1_bench_ndt_sort$sort.log
done
echo "== bench FAST_GICP (unchanged kernels, reference point)"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/v1_bench_gicp.log 2>&1; echo "exit $?"; tail -2 gpurun_out/v1_bench_gicp.log
echo "== odometry stream NDT"
timeout 600 python scripts/odometry_stream.py --help > gpurun_out/v1_odo_help.log 2>&1
timeout 900 python scripts/odometry_stream.py --method NDT_OMP --sensor HDL-64E --speed 3 --scans 30 > gpurun_out/v1_odo_ndt.log 2>&1; echo "exit $?"; tail -5 gpurun_out/v1_odo_ndt.log
echo "== rocprofv3 NDT bench"
(cd /tmp && HGS_BATCH_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/v1_prof_ndt" -o bench -- python "$OLDPWD/bench.py" --method NDT_OMP --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/v1_prof_ndt.log" 2>&1); echo "prof exit $?"
f=$(find gpurun_out/v1_prof_ndt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
