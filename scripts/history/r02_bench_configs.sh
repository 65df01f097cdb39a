#!/bin/bash
# Round 2, GPU visit 9: every BASELINE configuration as a bench.py line (committed under profiles/r02_bench_*.json)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout 900 python bench.py "$@" > gpurun_out/v9_$name.log 2>&1; echo "exit $?"; grep '^{' gpurun_out/v9_$name.log | tail -1 > gpurun_out/r02_bench_$name.json; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02_bench_$name.json"))
    print("  value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "timed_s", d.get("timed_region_s"), "seeds", d.get("value_by_scene_seed"))
    print("  step_ms", d.get("step_ms"), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
except Exception as e:
    print("  parse failed", e); print(open("gpurun_out/v9_$name.log").read()[-1500:])
PY
}
run metric_gicp
run metric_ndt --method NDT_OMP
run config2 --config 2
run config3 --config 3
run config4 --config 4
run config5 --config 5
run driver_cmd --gpus 1 --steps 20 --warmup 5
