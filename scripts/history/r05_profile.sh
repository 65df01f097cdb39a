#!/bin/bash
# Round 5 profile visit over the default bench workload (HDL-64E, 64 candidates per step) for the three lines the driver's JSON carries:
# FAST_GICP (FROBENIUS), FAST_GICP with PLANE regularisation, NDT_OMP.  Per line:
#   rocprofv3 --kernel-trace --stats, whole-device launches (HGS_BATCH_LANES=1)   -> gpurun_out/r05_<tag>_kernel_stats.md
#   the same in the TIMED configuration (default lanes)                           -> gpurun_out/r05_<tag>_kernel_stats_lanes.md
#   separate rocprofv3 --pmc passes (never combined with tracing), one lane        -> gpurun_out/pmc_<tag>/summary.{md,json}
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1"
for LINE in ${LINES:-fast_gicp fast_gicp_plane ndt_omp}; do
  case $LINE in
    fast_gicp) M="--method FAST_GICP";;
    fast_gicp_plane) M="--method FAST_GICP --regularization PLANE";;
    ndt_omp) M="--method NDT_OMP";;
  esac
  for MODE in one_lane lanes; do
    if [ $MODE = one_lane ]; then export HGS_BATCH_LANES=1; SUF=""; else unset HGS_BATCH_LANES; SUF="_lanes"; fi
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$LINE$SUF" -o bench -- python "$ROOT/bench.py" $M $ARGS > "$ROOT/gpurun_out/prof_$LINE$SUF.log" 2>&1); echo "trace $LINE $MODE exit $?"
    f=$(find gpurun_out/prof_$LINE$SUF -name "*kernel_stats.csv" | head -1)
    { echo "rocprofv3 --kernel-trace --stats -- python bench.py $M $ARGS   ($([ $MODE = one_lane ] && echo 'HGS_BATCH_LANES=1: whole-device launches' || echo 'default lanes: the configuration of the timed region'))"; echo; echo '```'; grep '^{' gpurun_out/prof_$LINE$SUF.log | tail -1 | cut -c1-2500; echo '```'; echo;
      [ -n "$f" ] && python scripts/prof_summary.py "$f"; } > gpurun_out/r05_${LINE}_kernel_stats$SUF.md
    head -9 gpurun_out/r05_${LINE}_kernel_stats$SUF.md | tail -4 | cut -c1-200
    find gpurun_out/prof_$LINE$SUF -name "*kernel_trace.csv" -delete
  done
  [ -n "${NO_PMC:-}" ] && continue
  export HGS_BATCH_LANES=1
  OUT="$ROOT/gpurun_out/pmc_$LINE"; mkdir -p "$OUT"
  run_pass() {
    local name="$1"; shift
    (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$ROOT/bench.py" $M $ARGS > "$OUT/$name.log" 2>&1)
    echo "pmc $LINE $name exit $?"
  }
  run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
  run_pass sq2 SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  run_pass sq3 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
  run_pass fetch FETCH_SIZE
  run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python scripts/pmc_summary.py "$OUT" "$OUT/summary.json" > "$OUT/summary.md"
  head -8 "$OUT/summary.md" | cut -c1-400
  find "$OUT" -name "*.csv" -delete
  unset HGS_BATCH_LANES
done
