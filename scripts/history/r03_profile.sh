#!/bin/bash
# Round 3 profile visit over the default bench workload (HDL-64E, 64 candidates per step), whole-device launches (HGS_BATCH_LANES=1):
#   rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/r03_<method>_kernel_stats.md
#   separate rocprofv3 --pmc passes (never combined with tracing)     -> gpurun_out/pmc_<method>/summary.{md,json}
# METHODS="FAST_GICP NDT_OMP" (default FAST_GICP); PRE="pytest args" runs a GPU test selection first.
set -u
export TMPDIR=/tmp
export HGS_BATCH_LANES=1
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
# LIB=ab_libs/<variant>.so profiles a library variant (scripts/build_variant.sh, scripts/pending/build_variants.sh) instead of the product library;
# TAG names its outputs (gpurun_out/r03_<method><TAG>_kernel_stats.md, pmc_<method><TAG>/)
if [ -n "${LIB:-}" ]; then cp "$LIB" hdl_graph_slam_amd/lib/libhgs_hip.so; fi
TAG="${TAG:-}"
if [ -n "${PRE:-}" ]; then echo "== pytest $PRE"; HGS_BATCH_LANES= timeout 900 python -m pytest $PRE -m gpu -q -x --timeout 400 -p no:cacheprovider 2>&1 | tail -4; fi
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ndt-record --seeds 1"
for M in ${METHODS:-FAST_GICP}; do
  m=$(echo $M | tr A-Z a-z)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$m" -o bench -- python "$ROOT/bench.py" --method $M $ARGS > "$ROOT/gpurun_out/prof_$m.log" 2>&1); echo "trace $M exit $?"
  f=$(find gpurun_out/prof_$m -name "*kernel_stats.csv" | head -1)
  { echo "rocprofv3 --kernel-trace --stats -- python bench.py --method $M $ARGS   (HGS_BATCH_LANES=1)"; echo; echo '```'; grep '^{' gpurun_out/prof_$m.log | tail -1; echo '```'; echo;
    [ -n "$f" ] && python scripts/prof_summary.py "$f"; } > gpurun_out/r03_${m}${TAG}_kernel_stats.md
  head -12 gpurun_out/r03_${m}${TAG}_kernel_stats.md | cut -c1-300
  [ -n "${NO_PMC:-}" ] && continue
  OUT="$ROOT/gpurun_out/pmc_$m$TAG"; mkdir -p "$OUT"
  run_pass() {
    local name="$1"; shift
    (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$ROOT/bench.py" --method $M $ARGS > "$OUT/$name.log" 2>&1)
    echo "pmc $M $name exit $?"
  }
  run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  run_pass sq2 SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  run_pass sq3 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
  run_pass fetch FETCH_SIZE
  run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python scripts/pmc_summary.py "$OUT" "$OUT/summary.json" > "$OUT/summary.md"
  head -8 "$OUT/summary.md" | cut -c1-600
  find "$OUT" -name "*.csv" -delete; find gpurun_out/prof_$m -name "*kernel_trace.csv" -delete
done
