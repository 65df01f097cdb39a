"""LoopDetector — Python mirror of include/hdl_graph_slam/loop_detector.hpp with the candidate loop
(loop_detector.hpp:135-154) replaced by ONE batched device call, optionally sharded over the GPUs of a node.

Semantics kept from the reference:
  * find_candidates: accumulated-distance and planar-distance gating (loop_detector.hpp:81-109);
  * guess = new_keyframe_estimate^-1 * candidate_estimate with both rotations re-normalised, guess(2,3) = 0 (:137-142);
  * per candidate: align + getFitnessScore(fitness_score_max_range) (:143-146);
  * selection: iterate in candidate order, skip if !converged or score > best_score, ties replace (:147-153);
  * reject if best_score > fitness_score_thresh (:160), else last_edge_accum_distance = new_keyframe.accum_distance (:168).
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .registration import RegistrationHIP, DeviceCloud, select_best
from .registrations import select_registration_method


@dataclasses.dataclass(eq=False)
class KeyFrame:
    """The fields of hdl_graph_slam::KeyFrame (include/hdl_graph_slam/keyframe.hpp:24-52) the loop detector reads."""
    cloud: np.ndarray                 # PointXYZI records
    estimate: np.ndarray              # node->estimate(): 4x4 float64
    accum_distance: float
    device_cloud: Optional[DeviceCloud] = None   # resident copy (uploaded once, reused for every later detection)
    id: int = -1                      # KeyFrame::id(): the pose-graph node id — the stable identity multi-GPU sharding uses


@dataclasses.dataclass(eq=False)
class Loop:
    key1: KeyFrame         # new keyframe (loop end)
    key2: KeyFrame         # matched candidate (loop begin)
    relative_pose: np.ndarray


def _normalized_rotation(T: np.ndarray) -> np.ndarray:
    """Eigen::Quaterniond(R).normalized().toRotationMatrix(): project onto SO(3) through the quaternion."""
    R = np.asarray(T, np.float64)[:3, :3]
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    w, x, y, z = q / np.linalg.norm(q)
    out = np.array(T, np.float64, copy=True)
    out[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return out


def loop_guess(new_keyframe_estimate: np.ndarray, candidate_estimate: np.ndarray) -> np.ndarray:
    """loop_detector.hpp:137-142."""
    a = _normalized_rotation(new_keyframe_estimate)
    b = _normalized_rotation(candidate_estimate)
    g = (np.linalg.inv(a) @ b).astype(np.float32)
    g[2, 3] = 0.0
    return g


class LoopDetector:
    def __init__(self, pnh, device_id: int = 0, registration: Optional[RegistrationHIP] = None, shard=None):
        get = pnh.get
        self.distance_thresh = float(get("distance_thresh", 5.0))                       # loop_detector.hpp:40
        self.accum_distance_thresh = float(get("accum_distance_thresh", 8.0))           # :41
        self.distance_from_last_edge_thresh = float(get("min_edge_interval", 5.0))      # :42
        self.fitness_score_max_range = float(get("fitness_score_max_range", L.DBL_MAX))  # :44
        self.fitness_score_thresh = float(get("fitness_score_thresh", 0.5))             # :45
        self.registration = registration or select_registration_method(pnh, device_id)  # :47
        self.last_edge_accum_distance = 0.0
        self.shard = shard  # optional hdl_graph_slam_amd.distributed.CandidateShard (multi-GPU)

    def detect(self, keyframes: Sequence[KeyFrame], new_keyframes: Sequence[KeyFrame]) -> List[Loop]:
        loops = []
        for nk in new_keyframes:
            loop = self.matching(self.find_candidates(keyframes, nk), nk)
            if loop is not None:
                loops.append(loop)
        return loops

    def find_candidates(self, keyframes: Sequence[KeyFrame], new_keyframe: KeyFrame) -> List[KeyFrame]:
        if new_keyframe.accum_distance - self.last_edge_accum_distance < self.distance_from_last_edge_thresh:
            return []
        out = []
        p2 = np.asarray(new_keyframe.estimate)[:2, 3]
        for k in keyframes:
            if new_keyframe.accum_distance - k.accum_distance < self.accum_distance_thresh:
                continue
            if np.linalg.norm(np.asarray(k.estimate)[:2, 3] - p2) > self.distance_thresh:
                continue
            out.append(k)
        return out

    def _resident(self, kf: KeyFrame) -> DeviceCloud:
        if kf.device_cloud is None:
            kf.device_cloud = self.registration.upload(kf.cloud)
        return kf.device_cloud

    def matching(self, candidates: Sequence[KeyFrame], new_keyframe: KeyFrame) -> Optional[Loop]:
        if not candidates:
            return None
        reg = self.registration
        reg.setInputTarget(self._resident(new_keyframe))                                  # loop_detector.hpp:122
        guesses = [loop_guess(new_keyframe.estimate, c.estimate) for c in candidates]
        if self.shard is not None:
            # shard by the keyframe's id, not by its position in this detection's list: the resident cloud (and its index /
            # covariances) of a keyframe then lives on one rank for good
            from .distributed import owner_of
            ids = [c.id if c.id >= 0 else i for i, c in enumerate(candidates)]
            mine = [owner_of(k, self.shard.world_size) == self.shard.rank for k in ids]
            records, best = self.shard.match(reg, [self._resident(c) if m else None for c, m in zip(candidates, mine)], guesses,
                                             self.fitness_score_max_range, keyframe_ids=ids)
        else:
            records, best = reg.loop_match_batch([self._resident(c) for c in candidates], guesses, self.fitness_score_max_range)
        self.last_records = records
        if best < 0:
            return None
        best_score = float(records["fitness_score"][best])
        if best_score > self.fitness_score_thresh:                                        # :160
            return None
        self.last_edge_accum_distance = new_keyframe.accum_distance                      # :168
        rel = np.array(records["final_transformation"][best], np.float32).reshape(4, 4).T.copy()
        return Loop(new_keyframe, candidates[best], rel)


def sequential_matching(reg, candidates, new_keyframe, fitness_score_max_range=L.DBL_MAX):
    """The reference's own loop (loop_detector.hpp:135-154) driven through the single-registration API; used by tests to
    check that the batched call selects the same candidate with the same pose."""
    reg.setInputTarget(new_keyframe.cloud)
    best_score, best, rel = L.DBL_MAX, -1, None
    for i, c in enumerate(candidates):
        reg.setInputSource(c.cloud)
        r = reg.align(loop_guess(new_keyframe.estimate, c.estimate))
        score = reg.getFitnessScore(fitness_score_max_range)
        if (not reg.hasConverged()) or score > best_score:
            continue
        best_score, best, rel = score, i, reg.getFinalTransformation()
    return best, best_score, rel
