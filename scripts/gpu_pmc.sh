#!/bin/bash
# PMC passes over a short bench run (one rocprofv3 invocation per counter set: --pmc is never combined with tracing).
# Output: gpurun_out/pmc/<pass>/..._counter_collection.csv ; summarised by scripts/pmc_summary.py
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc"
mkdir -p "$OUT"
ARGS="${BENCH_ARGS:---steps 1 --warmup 1 --no-cpu-baseline}"
cd /tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
run_pass() {
  local name="$1"; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run_pass sq2 SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*counter_collection.csv" | head
python "$ROOT/scripts/pmc_summary.py" "$OUT" "$OUT/summary.json" | tee "$OUT/summary.md"
