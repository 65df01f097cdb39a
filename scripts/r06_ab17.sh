#!/bin/bash
# Round 6, seventeenth A/B: the seed grid's point refined inside its own leaf (HGS_SEED_REFINE; ab_libs/norefine.so = without).  Fitness tests, the NDT_OMP
# metric batch with stage timers (k_fitness), calc_fitness_score.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
LOG=gpurun_out/r06_ab17.log
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
echo -n "tests (refined seeds): " | tee -a $LOG
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reference_code_pins.py tests/test_loop_detector.py -m gpu -x -q 2>&1 | tail -12 | grep -E "passed|failed|error" | tee -a $LOG
for rep in 1 2; do for v in norefine refine; do
  if [ "$v" = norefine ]; then cp ab_libs/norefine.so hdl_graph_slam_amd/lib/libhgs_hip.so; else cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so; fi
  echo -n "$v ndt: " | tee -a $LOG
  timeout 600 python bench.py --method NDT_OMP --steps 6 --warmup 2 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); s = r['roofline']['stage_ms_per_step']
    print(r['value'], r['ms_per_step'], 'fitness stage ms', s.get('fitness'), 'best', r.get('best_candidate'))
" | tee -a $LOG
done; done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
