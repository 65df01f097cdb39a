#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench line, rocprofv3 kernel stats.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
echo "== rocm-smi" > gpurun_out/env.log; rocm-smi --showproductname 2>&1 | head -20 >> gpurun_out/env.log; nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 400 --timeout-method=thread -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log
if [ -n "${DIAG:-}" ]; then echo "== diag"; timeout 1200 python scripts/gpu_diag.py $DIAG > gpurun_out/diag.log 2>&1; echo "diag exit $?"; cat gpurun_out/diag.log | tail -40; fi
echo "== bench"; timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench exit $?" | tee -a gpurun_out/bench.log; tail -3 gpurun_out/bench.log
if [ "${DO_PROF:-1}" = "1" ]; then
  # two kernel traces of the bench command: whole-device launches (engine option batch_lanes=1, the launch shape bench.py's HIP-event
  # roofline measurement times) and the default 4 concurrent lanes of the timed region
  for lanes in 1 4; do
    echo "== rocprofv3 lanes=$lanes"
    (cd /tmp && HGS_ENGINE_OPTIONS=batch_lanes=$lanes timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_lanes$lanes" -o bench -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OLDPWD/gpurun_out/prof_lanes$lanes.log" 2>&1); echo "prof exit $?"
    f=$(find gpurun_out/prof_lanes$lanes -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
  done
fi
