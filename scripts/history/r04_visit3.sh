#!/bin/bash
# Round 4, third GPU visit: what do the flush atomics of a single-cloud NDT pass cost (probe), and the A/B of this round's second batch of changes:
#   vH2 = the committed library; ndt_rep8 = + replicated NDT totals; lm_lds = + LM control step on LDS; greedy = + greedy bound for unseeded packets
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 60 scripts/probes/atomic_contention_probe 2>&1 | tee gpurun_out/r04_atomic_contention.log ) || echo "probe failed"
VARIANTS="vH2 ndt_rep8 lm_lds greedy" REPS=1 bash scripts/r04_ab.sh
VARIANTS="vH2 greedy" WORKLOADS="gicp ndt cfg2" REPS=1 bash scripts/r04_ab.sh
