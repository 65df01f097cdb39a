#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for l in 2 3 4; do echo "== NDT HGS_BATCH_LANES=$l"; HGS_BATCH_LANES=$l BENCH_FLAGS=" " STEPS=6 VARIANTS="v14" METHODS=NDT_OMP REPS=1 bash scripts/r03_ab.sh; done
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "small_trees or equidistant" -p no:cacheprovider 2>&1 | tail -2
