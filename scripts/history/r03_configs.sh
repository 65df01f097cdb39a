#!/bin/bash
# Round 3: the other BASELINE configurations on the current library (one JSON line each -> gpurun_out/r03_bench_config<N>.json)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for c in ${CONFIGS:-2 3 4 5}; do
  extra=""
  [ "$c" = 4 ] && extra="--fitness-max-range-variant"
  timeout 900 python bench.py --config $c ${FLAGS:-} $extra 2>gpurun_out/r03_bench_config$c.err | grep '^{' > gpurun_out/r03_bench_config$c.json
  python - "$c" <<'PY'
import json, sys
c = sys.argv[1]
try:
    r = json.load(open(f"gpurun_out/r03_bench_config{c}.json"))
    keys = ("value", "ms_per_step", "warm_value", "iterations", "mean_iterations", "latency_ms", "us_per_iteration_p50", "value_by_scene_seed")
    print(f"config {c}:", {k: r[k] for k in keys if k in r}, r["roofline"]["stage_ms_per_step"])
    for k in ("at_3_mps", "fitness_score_max_range_4", "oracle_stream", "trajectory_error_vs_ground_truth"):
        if k in r: print("   ", k, r[k])
    if r.get("cpu_baseline"): print("    cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"].get("single_thread"))
except Exception as e:
    print("config", c, "failed", e)
PY
done
if [ -n "${QPW_AB:-}" ]; then
  for q in 64 32 16; do echo -n "config 2 HGS_NN_QPW=$q: "; HGS_NN_QPW=$q timeout 300 python bench.py --config 2 --no-cpu-baseline --seeds 1 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        r = json.loads(ln); print(r['value'], r['ms_per_step'], r['warm_value'], r['roofline']['stage_ms_per_step'])"; done
fi
