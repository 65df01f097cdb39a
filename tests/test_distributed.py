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


class _StubEngine:
    """loop_match_batch of a registration stub: the record of a 'cloud' (an int: the keyframe id it was uploaded for) carries
    that id, and the stub remembers which keyframes it was asked to match."""
    def __init__(self):
        self.seen = []

    def loop_match_batch(self, clouds, guesses, max_range):
        from hdl_graph_slam_amd import _lib as L
        rec = np.zeros(len(clouds), dtype=L.RESULT_DTYPE)
        rec["converged"] = 1
        rec["iterations"] = np.asarray(clouds, np.int32) if len(clouds) else 0
        rec["fitness_score"] = [1.0 / (1 + k) for k in clouds]
        self.seen.append(list(clouds))
        return rec, -1


def _two_detections_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hdl_graph_slam_amd.distributed import CandidateShard, owner_of
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    shard, eng = CandidateShard(), _StubEngine()
    out = []
    for keyframe_ids in ([10, 11, 12, 13, 14], [14, 12, 10, 17]):            # the second detection lists them in another order
        cands = [k if owner_of(k, world) == rank else None for k in keyframe_ids]
        rec, best = shard.match(eng, cands, [np.eye(4)] * len(cands), 1.0, keyframe_ids=keyframe_ids)
        out.append((list(rec["iterations"]), best))
    q.put((rank, eng.seen, out))
    dist.destroy_process_group()


def test_sharding_follows_the_keyframe_id_not_the_list_position():
    """Two detections whose candidate lists order the same keyframes differently: a keyframe is matched by the rank
    of its id in both, so its resident cloud never has to move, and every rank still gets all records in list order."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_detections_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, out in res:
        assert all(k % world == rank for batch in seen for k in batch)        # only its own keyframes, in both detections
        assert out[0][0] == [10, 11, 12, 13, 14] and out[1][0] == [14, 12, 10, 17]   # records in the detection's list order
        assert out[0][1] == 4 and out[1][1] == 3      # lowest fitness score = largest id: position 4 of the first list, 3 of the second


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


def _sharded_equals_unsharded(make_engine, world_one_only=True):
    """hgs_comm_init + hgs_loop_match_batch_sharded at world size 1: the records gathered through the C entry point equal
    hgs_loop_match_batch's bit for bit, in candidate order, whatever order the rank lists its candidates in."""
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registration import RegistrationHIP
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=3, n_candidates=5, n_distinct=3, downsample=0.3)
    reg = make_engine()
    reg.setInputTarget(wl.target)
    clouds = [reg.upload(c) for c in wl.candidates]
    rec, best = reg.loop_match_batch(clouds, wl.guesses, 4.0)
    reg.comm_init(0, 1, RegistrationHIP.comm_unique_id())
    order = [3, 0, 4, 1, 2]
    rec_s, best_s = reg.loop_match_batch_sharded([clouds[i] for i in order], order, [wl.guesses[i] for i in order], 5, 4.0)
    assert best_s == best and list(rec_s["candidate_id"]) == [0, 1, 2, 3, 4]
    for name in ("final_transformation", "fitness_score", "converged", "iterations", "num_inliers", "lm_tries", "error"):
        assert np.array_equal(rec_s[name], rec[name]), name
    # a rank without candidates still takes part; candidates nobody reports come back "not converged"
    rec_e, best_e = reg.loop_match_batch_sharded([], [], [], 3, 4.0)
    assert best_e == -1 and not rec_e["converged"].any() and list(rec_e["candidate_id"]) == [0, 1, 2]
    reg.comm_finalize()
    reg.close()


@pytest.mark.gpu
def test_sharded_batch_through_the_c_abi_over_rccl():
    """World size 1 on the GPU: ncclCommInitRank + ncclAllGather on the engine's stream, no torch involved."""
    from hdl_graph_slam_amd.registrations import select_registration_method
    _sharded_equals_unsharded(lambda: select_registration_method({"registration_method": "FAST_GICP"}, device_id=0))


def _rccl_rank(rank, world, uid, n_total, q):
    import numpy as np
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.distributed import owner_of
    from hdl_graph_slam_amd.registrations import select_registration_method
    try:
        wl = workloads.make_loop_closure_set("VLP-16", scene_seed=3, n_candidates=n_total, n_distinct=3, downsample=0.3)
        reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
        reg.setInputTarget(wl.target)
        reg.comm_init(rank, world, uid)
        mine = [i for i in range(n_total) if owner_of(i, world) == rank]
        rec, best = reg.loop_match_batch_sharded([reg.upload(wl.candidates[i]) for i in mine], mine, [wl.guesses[i] for i in mine], n_total, 4.0)
        q.put((rank, "ok", rec.tobytes(), best))
        reg.close()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, "error", str(exc), -1))


@pytest.mark.gpu
def test_sharded_batch_two_ranks_on_one_gpu():
    """Two processes, two RCCL ranks, both on device 0 (the only GPU of the test box): every rank ends with all records in
    candidate order, equal to the unsharded batch.  RCCL refuses two ranks on one device in some configurations — then this
    test says so and the two-rank exchange stays covered by the gloo test above and the driver's multi-GPU bench."""
    import multiprocessing as mp
    from hdl_graph_slam_amd import workloads
    from hdl_graph_slam_amd.registration import RegistrationHIP
    from hdl_graph_slam_amd.registrations import select_registration_method
    n_total, world = 5, 2
    uid = RegistrationHIP.comm_unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_rank, args=(r, world, uid, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=240) for _ in procs]
    except Exception:  # noqa: BLE001  (a hung communicator: RCCL waiting for a second device)
        for p in procs:
            p.kill()
        pytest.skip("RCCL did not bring up two ranks on one device within 240 s")
    for p in procs:
        p.join(timeout=60)
    if any(r[1] != "ok" for r in res):
        pytest.skip(f"RCCL refuses two ranks on one device here: {[r[2] for r in res if r[1] != 'ok'][0][:200]}")
    wl = workloads.make_loop_closure_set("VLP-16", scene_seed=3, n_candidates=n_total, n_distinct=3, downsample=0.3)
    one = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
    one.setInputTarget(wl.target)
    rec, best = one.loop_match_batch([one.upload(c) for c in wl.candidates], wl.guesses, 4.0)
    rec["candidate_id"] = np.arange(n_total)
    for rank, _, blob, b in res:
        assert blob == rec.tobytes() and b == best
    one.close()


def test_merge_of_gathered_shard_records_is_plain_host_code():
    """hgs_debug_merge_shard_records — the merge step of hgs_loop_match_batch_sharded (hgs_engine.hip), replacing the loop of
    loop_detector.hpp:135-154 on a multi-GPU node — called in the REAL library on this CPU-only box: three ranks with uneven
    shards, padding slots, a rank that announced records but sent padding (it failed after the header), an out-of-range id and a
    candidate two ranks report."""
    import ctypes as C
    from hdl_graph_slam_amd import _lib as L
    lib = L.lib()
    world, per, n_total = 3, 3, 7
    g = np.zeros(world * per, dtype=L.RESULT_DTYPE)
    g["candidate_id"] = -1
    g["fitness_score"] = L.DBL_MAX

    def put(rank, slot, cid, fit, conv=1):
        r = g[rank * per + slot]
        r["candidate_id"], r["fitness_score"], r["converged"], r["iterations"] = cid, fit, conv, 3 + cid
    put(0, 0, 0, 0.5), put(0, 1, 3, 0.2), put(0, 2, 6, 0.9)      # rank 0: three records
    put(1, 0, 1, 0.3), put(1, 1, 4, 0.2, conv=0)                 # rank 1: two records + one padding slot
    put(2, 1, 5, 0.1)                                            # rank 2 announced 2 records: slot 0 is padding (it failed), slot 1 is real
    counts = np.array([3, 2, 2], np.int32)
    out = np.zeros(n_total, dtype=L.RESULT_DTYPE)
    dup = C.c_int32(123)
    vp = C.c_void_p
    assert lib.hgs_debug_merge_shard_records(g.ctypes.data_as(vp), counts.ctypes.data_as(vp), world, per, n_total, out.ctypes.data_as(vp), C.byref(dup)) == L.HGS_OK
    assert dup.value == -1 and list(out["candidate_id"]) == list(range(n_total))
    assert list(out["converged"]) == [1, 1, 0, 1, 0, 1, 1]                      # 2: nobody reported it; 4: reported, not converged
    assert out["fitness_score"][2] == L.DBL_MAX and out["iterations"][2] == 0
    assert list(out["fitness_score"][[0, 1, 3, 5, 6]]) == [0.5, 0.3, 0.2, 0.1, 0.9]
    best = C.c_int32(-1)
    assert lib.hgs_select_best(out.ctypes.data_as(vp), n_total, C.byref(best)) == L.HGS_OK and best.value == 5
    # slots beyond a rank's count are ignored even when they carry an id; ids outside [0, n_total) are dropped
    put(1, 2, 2, 0.01)
    put(2, 0, 99, 0.0)
    assert lib.hgs_debug_merge_shard_records(g.ctypes.data_as(vp), counts.ctypes.data_as(vp), world, per, n_total, out.ctypes.data_as(vp), C.byref(dup)) == L.HGS_OK
    assert dup.value == -1 and not out["converged"][2]
    # two ranks report candidate 3: flagged, first report kept
    put(2, 0, 3, 0.7)
    assert lib.hgs_debug_merge_shard_records(g.ctypes.data_as(vp), counts.ctypes.data_as(vp), world, per, n_total, out.ctypes.data_as(vp), C.byref(dup)) == L.HGS_OK
    assert dup.value == 3 and out["fitness_score"][3] == 0.2
    assert lib.hgs_debug_merge_shard_records(None, counts.ctypes.data_as(vp), world, per, n_total, out.ctypes.data_as(vp), C.byref(dup)) == L.HGS_ERR_INVALID_ARGUMENT
