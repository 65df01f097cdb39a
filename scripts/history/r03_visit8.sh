#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
VARIANTS="v11 v9 v10" REPS=2 bash scripts/r03_ab.sh
for l in 1 2 3 4; do echo "== HGS_BATCH_LANES=$l"; HGS_BATCH_LANES=$l VARIANTS="v11" REPS=1 bash scripts/r03_ab.sh; done
