"""Multi-GPU loop-closure matching: candidate keyframes are sharded over the ranks of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path has exactly one exchange step: after every rank has registered its own candidates against the
replicated target, the fixed-size per-candidate records are all-gathered and every rank applies the sequential selection
rule of loop_detector.hpp:146-153 to the full list.  Payload: 112 B x n_candidates (56 KiB for 512) — latency bound."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .registration import select_best


def owner_of(keyframe_id: int, world_size: int) -> int:
    """Interleaved partition by a STABLE identity: keyframe k -> rank k mod W.  A keyframe's position in a detection's
    candidate list changes from detection to detection, its id (the pose-graph node id, KeyFrame::id()) does not — so its
    resident cloud, search index and covariances stay on one GPU (adapters/loop_match_hip.hpp shards the same way).  Callers
    that have no ids (bench.py: every detection lists the same candidates) pass the list position."""
    return keyframe_id % world_size


class CandidateShard:
    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None, device=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self.device = device

    def owns(self, candidate_index: int) -> bool:
        return owner_of(candidate_index, self.world_size) == self.rank

    def local_indices(self, n_candidates: int):
        return [i for i in range(n_candidates) if self.owns(i)]

    def gather_records(self, local_records: np.ndarray, n_candidates: int) -> np.ndarray:
        """all-gather of per-candidate hgs_result records; returns them in global candidate order."""
        import torch
        W = self.world_size
        per_rank = (n_candidates + W - 1) // W
        rec_bytes = L.RESULT_DTYPE.itemsize
        send = np.zeros(per_rank, dtype=L.RESULT_DTYPE)
        send["candidate_id"] = -1
        send[: len(local_records)] = local_records
        t_send = torch.from_numpy(send.view(np.uint8).reshape(per_rank * rec_bytes).copy())
        if self.device is not None:
            t_send = t_send.to(self.device)
        t_recv = torch.empty(W * per_rank * rec_bytes, dtype=torch.uint8, device=t_send.device)
        self.dist.all_gather_into_tensor(t_recv, t_send, group=self.group)
        allrec = t_recv.cpu().numpy().view(L.RESULT_DTYPE).reshape(W, per_rank)
        out = np.zeros(n_candidates, dtype=L.RESULT_DTYPE)
        for c in range(n_candidates):
            r = allrec[owner_of(c, W), c // W]
            assert int(r["candidate_id"]) == c, "record / candidate mismatch after all-gather"
            out[c] = r
        return out

    def match(self, reg, candidates: Sequence, guesses: Sequence[np.ndarray], max_range: float, keyframe_ids: Optional[Sequence[int]] = None):
        """candidates[i] is a DeviceCloud on the owning rank (None elsewhere); keyframe_ids[i] its stable identity (default: i).
        Returns (records of ALL candidates, best).  The exchange goes through torch.distributed (gloo in the CPU tests)."""
        n = len(candidates)
        ids = list(range(n)) if keyframe_ids is None else list(keyframe_ids)
        mine = [i for i in range(n) if owner_of(ids[i], self.world_size) == self.rank]
        local, _ = reg.loop_match_batch([candidates[i] for i in mine], [guesses[i] for i in mine], max_range)
        local["candidate_id"] = np.asarray(mine, np.int32)
        records = self.gather_records_any(local, n)
        return records, select_best(records)

    def gather_records_any(self, local_records: np.ndarray, n_candidates: int) -> np.ndarray:
        """all-gather for an arbitrary partition: every rank sends n_candidates slots (its records, then candidate_id -1)."""
        import torch
        W, rec_bytes = self.world_size, L.RESULT_DTYPE.itemsize
        send = np.zeros(n_candidates, dtype=L.RESULT_DTYPE)
        send["candidate_id"] = -1
        send[: len(local_records)] = local_records
        t_send = torch.from_numpy(send.view(np.uint8).reshape(n_candidates * rec_bytes).copy())
        if self.device is not None:
            t_send = t_send.to(self.device)
        t_recv = torch.empty(W * n_candidates * rec_bytes, dtype=torch.uint8, device=t_send.device)
        self.dist.all_gather_into_tensor(t_recv, t_send, group=self.group)
        allrec = t_recv.cpu().numpy().view(L.RESULT_DTYPE)
        out = np.zeros(n_candidates, dtype=L.RESULT_DTYPE)
        out["candidate_id"] = np.arange(n_candidates)
        out["fitness_score"] = L.DBL_MAX
        for r in allrec:
            c = int(r["candidate_id"])
            if 0 <= c < n_candidates:
                out[c] = r
        return out

    def match_rccl(self, reg, candidates: Sequence, guesses: Sequence[np.ndarray], max_range: float, keyframe_ids: Optional[Sequence[int]] = None):
        """The same exchange through the C-ABI (hgs_loop_match_batch_sharded): RCCL all-gather on the engine's own stream, no
        torch tensor, no host staging.  reg.comm_init(rank, world, id) must have been called (comm_bootstrap below)."""
        n = len(candidates)
        ids = list(range(n)) if keyframe_ids is None else list(keyframe_ids)
        mine = [i for i in range(n) if owner_of(ids[i], self.world_size) == self.rank]
        return reg.loop_match_batch_sharded([candidates[i] for i in mine], mine, [guesses[i] for i in mine], n, max_range)

    def comm_bootstrap(self, reg):
        """hgs_comm_init on every rank: rank 0's RCCL id travels through the existing torch.distributed group."""
        from .registration import RegistrationHIP
        box = [RegistrationHIP.comm_unique_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        reg.comm_init(self.rank, self.world_size, box[0])


class MultiDeviceLoopMatcher:
    """Single-process variant for hosts that are one process by construction (the ROS nodelet manager): one engine handle
    per local GPU, candidate c -> engine c mod N (the partition of CandidateShard), every shard matched from its own host
    thread (hgs handles are independent: own stream, own buffers), results merged on the host with the sequential selection
    rule.  No collective is needed: the per-candidate records already live in host memory of the one process."""

    def __init__(self, pnh: dict, device_ids):
        from .registrations import select_registration_method
        self.engines = [select_registration_method(pnh, device_id=d) for d in device_ids]

    def close(self):
        for e in self.engines:
            e.close()

    def upload(self, keyframe_id: int, cloud):
        """Keyframe clouds live on the device of the engine that will match them: the engine of keyframe_id mod N, whatever
        position the keyframe takes in a later detection's candidate list."""
        return self.engines[owner_of(keyframe_id, len(self.engines))].upload(cloud)

    def match(self, target, candidates: Sequence, guesses: Sequence[np.ndarray], max_range: float = L.DBL_MAX, keyframe_ids: Optional[Sequence[int]] = None):
        """candidates[i]: the DeviceCloud upload(keyframe_ids[i], ...) returned (keyframe_ids default to the list positions).
        Returns (records of all candidates in list order, best)."""
        import threading
        n, W = len(candidates), len(self.engines)
        ids = list(range(n)) if keyframe_ids is None else list(keyframe_ids)
        out = np.zeros(n, dtype=L.RESULT_DTYPE)
        errors = []

        def work(r):
            try:
                mine = [i for i in range(n) if owner_of(ids[i], W) == r]
                if not mine:
                    return
                eng = self.engines[r]
                eng.setInputTarget(target)       # host cloud: every engine builds its own copy (cheaper than a broadcast)
                rec, _ = eng.loop_match_batch([candidates[i] for i in mine], [guesses[i] for i in mine], max_range)
                rec["candidate_id"] = np.asarray(mine, np.int32)
                out[mine] = rec
            except Exception as exc:   # surfaced after join
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(r,)) for r in range(W)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out, select_best(out)
