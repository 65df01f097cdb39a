#!/bin/bash
# (record of the command: the variants it names were built from round 3's scripts/pending patches, which this visit measured and the round removed)
# Round 4, first GPU visit: same-box A/B of scripts/pending (round 3's unmeasured exact reductions) plus the LDS lane-slot reduction written
# this round, the MFMA wave sums of k_gicp_linearize, and the grid-barrier probe that decides the single-registration design.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 60 scripts/probes/grid_barrier_probe 2>&1 | tee gpurun_out/r04_grid_barrier.log ) || echo "probe failed / timed out"
VARIANTS="vH ndt_pass_items32 ndt_reduction_scalar_scales ndt_reduction_mfma ndt_lds ndt_lds2" TEST_TIMEOUT=170 bash scripts/r03_ab_ndt.sh
VARIANTS="vH gicp_linearize_sums_mfma" BENCH_FLAGS="--distinct 4" REPS=2 bash scripts/r03_ab.sh
