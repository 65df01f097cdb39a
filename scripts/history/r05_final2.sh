#!/bin/bash
# Round 5, the closing visit on the final library (after the campaign of r05_final.sh the upload / download paths, k_bbox_count, the comm_wait backoff and bench.py's
# queue default changed): the whole -m gpu suite + smoke, the driver's command, configs 2 / 3 / 5 and the process-group / single-process records again.
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
SKIP_PROFILE=1 SKIP_CONFIGS=1 bash scripts/r05_final.sh
CONFIGS="2 3 5" bash scripts/r05_configs.sh
