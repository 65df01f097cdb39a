#!/bin/bash
# Round 2, GPU visit 3: hardware counters of k_ndt_pass on the NDT loop batch (whole-device launches).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/v3_pmc"
mkdir -p "$OUT"
export HGS_BATCH_LANES=1
ARGS="--method NDT_OMP --steps 1 --warmup 1 --no-cpu-baseline --candidates 8 --distinct 4"
cd /tmp
run_pass() {
  local name="$1"; shift
  timeout 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run_pass sq2 SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run_pass sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_GDS
python "$ROOT/scripts/pmc_summary.py" "$OUT" "$OUT/summary.json" | tee "$OUT/summary.md" | head -60
