#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for m in 1 2 1 2; do echo "== HGS_KNN_REPLAY=$m"; HGS_KNN_REPLAY=$m VARIANTS="v8" REPS=1 bash scripts/r03_ab.sh; done
for m in 0 2; do echo "== config 2/5 HGS_KNN_REPLAY=$m"; HGS_KNN_REPLAY=$m CONFIGS="2 5" FLAGS="--no-cpu-baseline --seeds 1" bash scripts/r03_configs.sh; done
