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


def owner_of(candidate_index: int, world_size: int) -> int:
    """Interleaved partition: candidate c -> rank c mod W (balances distance-sorted candidate lists)."""
    return candidate_index % world_size


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

    def match(self, reg, candidates: Sequence, guesses: Sequence[np.ndarray], max_range: float):
        """candidates[i] is a DeviceCloud on the owning rank (None elsewhere). Returns (records of ALL candidates, best)."""
        n = len(candidates)
        mine = self.local_indices(n)
        local, _ = reg.loop_match_batch([candidates[i] for i in mine], [guesses[i] for i in mine], max_range)
        local["candidate_id"] = np.asarray(mine, np.int32)
        records = self.gather_records(local, n)
        return records, select_best(records)


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

    def upload(self, candidate_index: int, cloud):
        """Keyframe clouds live on the device of the engine that will match them."""
        return self.engines[owner_of(candidate_index, len(self.engines))].upload(cloud)

    def match(self, target, candidates: Sequence, guesses: Sequence[np.ndarray], max_range: float = L.DBL_MAX):
        """candidates[i]: DeviceCloud created by upload(i, ...). Returns (records of all candidates in order, best)."""
        import threading
        n, W = len(candidates), len(self.engines)
        out = np.zeros(n, dtype=L.RESULT_DTYPE)
        errors = []

        def work(r):
            try:
                mine = [i for i in range(n) if owner_of(i, W) == r]
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
