#!/bin/bash
# Round 2 profile visit over the default bench workload (HDL-64E, 64 candidates per step), whole-device launches
# (HGS_BATCH_LANES=1: the launch shape bench.py's HIP-event roofline measurement times):
#   1. rocprofv3 --kernel-trace --stats of the bench command, FAST_GICP and NDT_OMP  -> gpurun_out/r02_<method>_kernel_stats.md
#   2. separate rocprofv3 --pmc passes (never combined with tracing)                  -> gpurun_out/pmc_<method>/summary.{md,json}
# Copy the summaries into profiles/ afterwards.
set -u
export TMPDIR=/tmp
export HGS_BATCH_LANES=1
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --seeds 1"
echo "== ndt edge cases"; timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "ndt_edge" -p no:cacheprovider 2>&1 | tail -3
for M in FAST_GICP NDT_OMP; do
  m=$(echo $M | tr A-Z a-z)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$m" -o bench -- python "$ROOT/bench.py" --method $M $ARGS > "$ROOT/gpurun_out/prof_$m.log" 2>&1); echo "trace $M exit $?"
  f=$(find gpurun_out/prof_$m -name "*kernel_stats.csv" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python bench.py --method $M $ARGS   (HGS_BATCH_LANES=1)"; echo; echo '```'; grep '^{' gpurun_out/prof_$m.log | tail -1; echo '```'; echo;
    [ -n "$f" ] && python scripts/prof_summary.py "$f"; } > gpurun_out/r02_${m}_kernel_stats.md
  head -12 gpurun_out/r02_${m}_kernel_stats.md | cut -c1-300
  OUT="$ROOT/gpurun_out/pmc_$m"; mkdir -p "$OUT"
  run_pass() {
    local name="$1"; shift
    (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$ROOT/bench.py" --method $M $ARGS > "$OUT/$name.log" 2>&1)
    echo "pmc $M $name exit $?"
  }
  run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  run_pass sq2 SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  run_pass fetch FETCH_SIZE
  run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python scripts/pmc_summary.py "$OUT" "$OUT/summary.json" > "$OUT/summary.md"
  head -8 "$OUT/summary.md" | cut -c1-400
  # the raw per-dispatch CSVs are large: keep the summaries only
  find "$OUT" -name "*.csv" -delete; find gpurun_out/prof_$m -name "*kernel_trace.csv" -delete
done
