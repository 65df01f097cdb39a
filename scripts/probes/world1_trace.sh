#!/bin/bash
# kernel trace of the process-group path at world size 1 with the library's RCCL exchange: what do the two all-gathers cost on the device, and on the host?
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"; mkdir -p gpurun_out
[ -d .scan_cache ] && export HGS_SCAN_CACHE="$ROOT/.scan_cache"
A="--gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-ndt-record --no-plane-record --seeds 1"
(cd /tmp && HGS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/trace_w1" -o t -- python "$ROOT/bench.py" $A > "$ROOT/gpurun_out/trace_w1.log" 2>&1); echo "exit $?"
g=$(find gpurun_out/trace_w1 -name "*kernel_stats.csv" | head -1)
[ -n "$g" ] && python scripts/prof_summary.py "$g" | head -30
find gpurun_out/trace_w1 -name "*.csv" -delete
python - <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
from hdl_graph_slam_amd import workloads, _lib as L
from hdl_graph_slam_amd.registrations import select_registration_method
from hdl_graph_slam_amd.registration import RegistrationHIP
wl = workloads.make_loop_closure_set("VLP-16", 3, n_candidates=8, n_distinct=4, downsample=0.3)
reg = select_registration_method({"registration_method": "FAST_GICP"})
reg.setInputTarget(wl.target)
cl = [reg.upload(c) for c in wl.candidates]
reg.comm_init(0, 1, RegistrationHIP.comm_unique_id())
ids = np.arange(8, dtype=np.int32)
for name, f in (("hgs_loop_match_batch", lambda: reg.loop_match_batch(cl, wl.guesses, 4.0)), ("hgs_loop_match_batch_sharded (world 1)", lambda: reg.loop_match_batch_sharded(cl, ids, wl.guesses, 8, 4.0))):
    for _ in range(3): f()
    ts = []
    for _ in range(30):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    print(name, "p50 ms", round(float(np.median(ts)), 3), "min", round(min(ts), 3))
reg.close()
PY
