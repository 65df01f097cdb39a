#!/usr/bin/env python3
"""bench.py — registrations/sec of the MI355X scan-matching backend on BASELINE.json's metric.

Workload (config.workload): the loop-closure batch of LoopDetector::matching (include/hdl_graph_slam/loop_detector.hpp:117-171)
on 64-beam ~120 k-point keyframes: per step every rank registers `--candidates` candidate keyframes (cold: search index +
20-NN covariances are recomputed for the target and every candidate, as the reference does per setInputSource) against one
query keyframe with FAST_GICP at the launch-file parameters (launch/hdl_graph_slam.launch:73-82,127-136: eps 0.01,
max_iterations 64, max_correspondence_distance 2.5, k 20) and evaluates getFitnessScore for each.  Inputs are resident in HBM
before the timed region.  N > 1: candidates are sharded over the ranks (weak scaling: per-GPU work fixed) and the per-candidate
records are all-gathered over RCCL once per step (the path's only exchange step).

One JSON line on rank 0:  value = registrations/sec over all ranks (whole job)."""
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
    "covariance": ("k_knn_cov", 360.0),    # per point (16 query + 20*16 neighbours + 24 covariance out)
    "linearize": ("k_gicp_linearize", 84.0),  # per source point per linearisation
    "error": ("k_gicp_error", 84.0),       # per source point per LM trial
    "fitness": ("k_fitness", 32.0),        # per source point
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # MI355X_MICROARCH.md: measured float4-copy ceiling (SURVEY 8d asks for both)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--candidates", type=int, default=16, help="candidate keyframes per GPU per step")
    ap.add_argument("--sensor", default="HDL-64E")
    ap.add_argument("--distinct", type=int, default=8, help="distinct ray-cast scans behind the candidates")
    ap.add_argument("--method", default="FAST_GICP", choices=["FAST_GICP", "FAST_VGICP", "NDT_OMP"])
    ap.add_argument("--downsample", type=float, default=0.0, help="voxel size applied to every keyframe (0 = raw scans, the metric's configuration)")
    ap.add_argument("--ndt-line-search", action="store_true", help="NDT_OMP with the opt-in More-Thuente search (NOT the reference's behaviour; "
                    "the line then says so in config.workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=3, help="candidates registered by the CPU oracle for cpu_baseline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    pnh = {"registration_method": args.method}
    if args.method in ("NDT_OMP", "FAST_VGICP"):
        pnh["reg_resolution"] = 1.0   # launch files use 1.0 (NDT factory default 0.5)
    if args.ndt_line_search and args.method == "NDT_OMP":
        pnh["reg_ndt_line_search"] = True
    B = args.candidates
    # every rank holds the query keyframe (replicated target) and its own shard of the N*B candidates
    wl = workloads.make_loop_closure_set(args.sensor, scene_seed=0, n_candidates=B, n_distinct=min(args.distinct, B), downsample=args.downsample or None)
    rng = np.random.default_rng(100 + rank)
    if rank > 0:  # different guesses per rank so that the shards are not identical problems
        for g in wl.guesses:
            g[:3, 3] += rng.normal(0, 0.05, 3).astype(np.float32)
            g[2, 3] = 0.0

    reg = select_registration_method(pnh, device_id=local_rank)
    d_target = reg.upload(wl.target)
    d_cands = [reg.upload(c) for c in wl.candidates]
    n_pts = [len(c) for c in wl.candidates]
    shard = CandidateShard(rank, world, device=coll_device) if sharded else None

    def step(cold=True):
        d_target.invalidate()           # the query keyframe is new in every detection
        if cold:
            for c in d_cands:           # reference behaviour: setInputSource rebuilds tree + covariances of every candidate
                c.invalidate()
        reg.setInputTarget(d_target)
        rec, best = reg.loop_match_batch(d_cands, wl.guesses, L.DBL_MAX)
        if shard is not None:
            rec["candidate_id"] = np.arange(rank, world * B, world, dtype=np.int32)
            allrec = shard.gather_records(rec, world * B)
            best = select_best(allrec)
        return rec, best

    def barrier():
        reg.synchronize()
        if not emulated:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec, best = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- informational: the same batch with the candidate keyframes' index + covariances kept resident between
    # detections (what a keyframe device cache gives; never `value`)
    step(cold=False)
    barrier()
    tw = time.perf_counter()
    for _ in range(args.steps):
        step(cold=False)
    barrier()
    dt_warm = time.perf_counter() - tw

    # ---- accuracy of the timed results (rank-local): vs ground truth
    et = [synth.pose_error(np.array(r["final_transformation"]).reshape(4, 4).T, Tg) for r, Tg in zip(rec, wl.T_gt)]
    rmse_t = float(np.sqrt(np.mean([e[0] ** 2 for e in et])))
    rmse_r = float(np.sqrt(np.mean([e[1] ** 2 for e in et])))

    # ---- roofline of the dominant kernel: HIP events on the handle's stream around every launch of each stage
    reg.profile_enable(True)
    reg.profile_read(reset=True)
    prof_steps = 2
    for _ in range(prof_steps):
        rec_p, _ = step()
    prof = reg.profile_read(reset=True)
    reg.profile_enable(False)
    total_pts = sum(n_pts) + len(wl.target)
    units = {
        "covariance": prof_steps * total_pts,
        "linearize": prof_steps * float(np.sum(rec_p["iterations"].astype(np.float64) * np.array(n_pts))),
        "error": prof_steps * float(np.sum(rec_p["lm_tries"].astype(np.float64) * np.array(n_pts))),
        "fitness": prof_steps * sum(n_pts),
    }
    if args.method == "NDT_OMP":
        ALG_BYTES["linearize"] = ("k_ndt_pass", 296.0)
        units["linearize"] = prof_steps * float(np.sum(rec_p["lm_tries"].astype(np.float64) * np.array(n_pts)))
        units["covariance"] = units["error"] = 0.0
    if args.method == "FAST_VGICP":
        ALG_BYTES["linearize"] = ("k_vgicp_linearize", 80.0)
        ALG_BYTES["error"] = ("k_vgicp_error", 80.0)
    dom = max((s for s in ALG_BYTES if prof[s][1] > 0), key=lambda s: prof[s][0])
    ms, launches = prof[dom]
    kname, bytes_per_unit = ALG_BYTES[dom]
    achieved = units[dom] * bytes_per_unit / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM traffic of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per dispatch, collected in separate passes over this
    # same command (scripts/gpu_pmc.sh) and committed as profiles/pmc_<method>.json; FETCH_SIZE doubled as
    # MI355X_MICROARCH.md prescribes for gfx950 (calibrated there for wide coalesced reads only — our reads are 128-byte
    # records and 16-byte gathers, so treat it as an upper estimate).  null when no committed counters match.
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.method.lower()}.json")
    default_workload = args.sensor == "HDL-64E" and B == 16 and not args.downsample   # what the committed counters were collected on
    if os.path.exists(pmc_path) and default_workload:
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        for kn, cv in pmc.get("kernels", {}).items():
            if kn.split("<")[0].endswith(kname) and "FETCH_SIZE" in cv:
                traffic = (2.0 * cv["FETCH_SIZE"] + cv.get("WRITE_SIZE", 0.0)) * 1024.0
                traffic_src = os.path.relpath(pmc_path, ROOT)
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 5), "traffic": None if traffic is None else round(traffic, 1), "traffic_source": traffic_src,
                "limiter": {"FAST_GICP": "instruction issue (VALU+SALU) of the exact tree search, not HBM: see DESIGN.md section 4",
                            "FAST_VGICP": "VALU + dependent L2 round trips of the voxel hash probes, not HBM: see DESIGN.md section 4",
                            "NDT_OMP": "VALU issue of the per-cell derivative terms (packed fp32 + fp64 accumulation, ~300 instructions per visited cell), not HBM: see DESIGN.md section 4"}[args.method],
                "launch_config": "whole-device launches (all candidates of the rank in one launch, HIP events on the engine's stream); the timed region "
                                 "runs the same kernels split over 4 concurrent lanes, whose launches overlap each other",
                "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2), "launches": launches,
                "algorithmic_bytes_per_launch": round(units[dom] * bytes_per_unit / max(launches, 1), 1),
                "stage_ms_per_step": {s: round(prof[s][0] / prof_steps, 3) for s in prof}}

    # ---- CPU baseline: the oracle (port of fast_gicp / ndt_omp, OpenMP over points) on a bounded sample, rank 0, N == 1
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        p = O.HgsParams()
        for name, _ in O.HgsParams._fields_:
            setattr(p, name, getattr(reg.params, name))
        # reg_num_threads = 0 means "all cores" upstream; on many-core hosts that oversubscribes badly (256 threads are
        # ~40x slower than 32 on a 2 x EPYC 9575F box), so the baseline is the BEST of a few thread counts.
        ncpu = os.cpu_count() or 1
        k = min(args.cpu_sample, B)
        best_cpu = None
        for nt in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
            O.set_num_threads(nt)
            o = O.OracleRegistration(p)
            o.setInputTarget(wl.target)   # target structures are built once per batch in the reference too
            o.setInputSource(wl.candidates[0])
            o.align(wl.guesses[0])        # warm-up (first-touch, thread pool, target covariances)
            tc = time.perf_counter()
            dpose = []
            for i in range(k):
                o.setInputSource(wl.candidates[i])
                ro = o.align(wl.guesses[i])
                o.getFitnessScore()
                dpose.append(synth.pose_error(np.array(rec[i]["final_transformation"]).reshape(4, 4).T, ro.matrix()))
            tcpu = time.perf_counter() - tc
            if best_cpu is None or k / tcpu > best_cpu[0]:
                best_cpu = (k / tcpu, nt, dpose)
        cpu = {"value": round(best_cpu[0], 4), "unit": "registrations/sec", "cores": best_cpu[1], "kind": "port", "host_threads_available": ncpu,
               "sample": f"{k} of the {B} candidate registrations (setInputSource + align + getFitnessScore), target structures prebuilt; "
                         f"best of OMP thread counts 8/16/32/64",
               "max_pose_diff_vs_gpu_m": float(max(d[0] for d in best_cpu[2])), "max_pose_diff_vs_gpu_rad": float(max(d[1] for d in best_cpu[2]))}

    if rank == 0:
        regs = world * B * args.steps
        out = {
            "metric": "registrations/sec (64-beam ~120k-pt pair), loop-closure batch", "value": round(regs / dt, 3), "unit": "registrations/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.method == "NDT_OMP" else "f64",
            "data": "synthetic" if not emulated else "synthetic (EMULATED ON THE CPU - not a measurement)",
            "config": {"workload": f"loop-closure batch: {B} candidate keyframes/GPU x {args.sensor} (~{int(np.mean(n_pts))} pts) vs 1 query keyframe, "
                                   f"{args.method}{' with the opt-in More-Thuente line search (not the reference behaviour)' if pnh.get('reg_ndt_line_search') else ''}"
                                   f" + getFitnessScore, cold (index + covariances rebuilt every step)",
                       "candidates_per_gpu": B, "points_per_cloud": int(np.mean(n_pts)), "method": args.method,
                       "parallelism": f"candidate-sharded x{world}" if world > 1 else "single GPU"},
            "pose_rmse_vs_ground_truth": {"translation_m": round(rmse_t, 5), "rotation_rad": round(rmse_r, 6)},
            "resident_keyframes_value": round(world * B * args.steps / dt_warm, 3) if world == 1 else None,
            "converged": int(np.sum(rec["converged"])), "mean_iterations": float(np.mean(rec["iterations"])),
            "mean_linearizations": float(np.mean(rec["lm_tries"])), "best_candidate": int(best),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    reg.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
