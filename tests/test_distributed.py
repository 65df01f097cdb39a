"""Multi-GPU sharding of the loop-closure batch, exercised with world_size 2 on CPU (gloo): partition, all-gather of the
per-candidate records, identical selection on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_records(n, seed=0):
    from hdl_graph_slam_amd import _lib as L
    rng = np.random.default_rng(seed)
    rec = np.zeros(n, dtype=L.RESULT_DTYPE)
    rec["final_transformation"] = rng.normal(size=(n, 16)).astype(np.float32)
    rec["converged"] = rng.random(n) > 0.2
    rec["fitness_score"] = np.round(rng.random(n), 1)       # coarse -> ties exist
    rec["iterations"] = rng.integers(1, 20, n)
    rec["candidate_id"] = np.arange(n)
    return rec


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hdl_graph_slam_amd.distributed import CandidateShard
    from hdl_graph_slam_amd.registration import select_best
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    full = _fake_records(n)
    shard = CandidateShard()
    mine = shard.local_indices(n)
    assert all(shard.owns(i) for i in mine)
    gathered = shard.gather_records(full[mine].copy(), n)
    q.put((rank, gathered.tobytes(), select_best(gathered), mine))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8, 1])
def test_all_gather_of_candidate_records_world2(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from hdl_graph_slam_amd.registration import select_best
    full = _fake_records(n)
    owned = sorted(i for r in res for i in r[3])
    assert owned == list(range(n))                                   # a partition: every candidate exactly once
    for rank, blob, best, mine in res:
        assert blob == full.tobytes()                                # every rank sees all records in candidate order
        assert best == select_best(full)                             # and selects the same candidate as the sequential rule


def test_owner_is_interleaved():
    from hdl_graph_slam_amd.distributed import owner_of
    assert [owner_of(c, 4) for c in range(9)] == [0, 1, 2, 3, 0, 1, 2, 3, 0]


@pytest.mark.gpu
def test_multi_device_matcher_equals_single_batch():
    """The single-process multi-GPU path (one handle per device, host threads) on two handles of device 0 gives the records
    of one batch on one handle, candidate for candidate."""
    sys.path.insert(0, ROOT)
    from hdl_graph_slam_amd import workloads, _lib as L
    from hdl_graph_slam_amd.distributed import MultiDeviceLoopMatcher
    from hdl_graph_slam_amd.registrations import select_registration_method
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=3, n_candidates=5, n_distinct=3, downsample=0.2)
    pnh = {"registration_method": "FAST_GICP"}
    mm = MultiDeviceLoopMatcher(pnh, [0, 0])
    cands = [mm.upload(i, c) for i, c in enumerate(wl.candidates)]
    rec_m, best_m = mm.match(wl.target, cands, wl.guesses)
    one = select_registration_method(pnh, device_id=0)
    one.setInputTarget(wl.target)
    rec_1, best_1 = one.loop_match_batch([one.upload(c) for c in wl.candidates], wl.guesses)
    assert best_m == best_1
    for a, b in zip(rec_m, rec_1):
        assert a["converged"] == b["converged"] and a["iterations"] == b["iterations"]
        assert np.abs(np.array(a["final_transformation"]) - np.array(b["final_transformation"])).max() < 1e-6
        assert abs(a["fitness_score"] - b["fitness_score"]) <= 1e-9 * abs(b["fitness_score"])
    mm.close(), one.close()


@pytest.mark.gpu
def test_bench_process_group_path_over_rccl():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, backend nccl = RCCL), with one
    rank: the all-gather of the candidate records, the max-over-ranks all-reduce and the barriers run on device tensors."""
    import json
    import subprocess
    env = dict(os.environ, HGS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--candidates", "4", "--sensor", "HDL-32E", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["converged"] == 4 and rec["roofline"]["frac"] > 0
