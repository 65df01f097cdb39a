#!/bin/bash
# Round 5: which LDS access of k_ndt_pass meets busy banks?  Suspect by the banking rules: ndt_flush's rotated reads / clearing writes.
# Same-box A/B of the round-4 rotation (ab_libs/ndt_rot0.so) against the conflict-free one (ab_libs/ndt_rot1.so): the NDT batch's rate, and one rocprofv3 --pmc
# pass each (counters only — never combined with tracing) over the same command.
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
cp hdl_graph_slam_amd/lib/libhgs_hip.so /tmp/current.so
ARGS="--method NDT_OMP --steps 8 --warmup 2 --no-cpu-baseline --no-ndt-record --seeds 1"
for rep in 1 2 3; do for v in ndt_rot0 ndt_rot1; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  echo -n "$v: "
  timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], 'p50', r['step_ms']['p50'], 'its', r['mean_iterations'], 'conv', r['converged'], 'best', r['best_candidate'], 'pass us', r['roofline']['avg_launch_us'])"
done; done 2>&1 | tee gpurun_out/r05_ndt_lds.log
for v in ndt_rot0 ndt_rot1; do
  cp ab_libs/$v.so hdl_graph_slam_amd/lib/libhgs_hip.so
  OUT="$ROOT/gpurun_out/pmc_$v"; rm -rf "$OUT"; mkdir -p "$OUT"
  (cd /tmp && HGS_BATCH_LANES=1 timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq" -o pmc -- python "$ROOT/bench.py" --method NDT_OMP --steps 3 --warmup 1 --no-cpu-baseline --no-ndt-record --seeds 1 > "$OUT/sq.log" 2>&1)
  echo "pmc $v exit $?" | tee -a gpurun_out/r05_ndt_lds.log
  python scripts/pmc_summary.py "$OUT" "$OUT/summary.json" > "$OUT/summary.md" 2>> gpurun_out/r05_ndt_lds.log
  grep -E "kernel|k_ndt_pass" "$OUT/summary.md" | cut -c1-600 | tee -a gpurun_out/r05_ndt_lds.log
  find "$OUT" -name "*.csv" -delete
done
cp /tmp/current.so hdl_graph_slam_amd/lib/libhgs_hip.so
