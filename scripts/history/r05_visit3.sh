#!/bin/bash
# Round 5, visit 3: the host-path latency changes (pinned upload ring instead of staging syncs, upload waits for the copy only, meta reset folded into
# the packing kernel, independent loads in k_bbox_count, accumulator zeroing folded into k_ndt_init) — same-box A/B on the single-registration
# configurations, then the whole -m gpu suite on the new library.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
VARIANTS="base lat" WORKLOADS="cfg3 cfg2 gicp" REPS=2 bash scripts/r05_ab.sh
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$PWD/.scan_cache"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r05_v3_tests.log
