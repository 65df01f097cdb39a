"""Host mirror of hdl_graph_slam::InformationMatrixCalculator (include/hdl_graph_slam/information_matrix_calculator.hpp:14-59,
src/hdl_graph_slam/information_matrix_calculator.cpp:10-47): the information matrix of an odometry / loop edge from the fitness score of the two
keyframe clouds.  The fitness score itself is the device kernel (`RegistrationHIP.calc_fitness_score` -> `hgs_calc_fitness_score`, row f1); what is
left on the host is a dozen scalar operations, restated here with the reference's quirks: the weights are truncated to `float` before the
division (:40-41), the constructor's `fitness_score_thresh` default is 0.5 while `load()`'s is 2.5 (hpp:32), and the constant-matrix branch
divides by the standard deviation, not the variance (:28-29).  Pinned against the reference's own translation unit run in the build container
(tests/test_reference_code_pins.py)."""
from __future__ import annotations

import math

import numpy as np

_DEFAULTS = dict(use_const_inf_matrix=False, const_stddev_x=0.5, const_stddev_q=0.1, var_gain_a=20.0, min_stddev_x=0.1, max_stddev_x=5.0,
                 min_stddev_q=0.05, max_stddev_q=0.2, fitness_score_thresh=0.5)


class InformationMatrixCalculator:
    def __init__(self, params: dict | None = None, loaded: bool = False):
        """`params`: rosparams (constructor, .cpp:10-21).  `loaded=True` = the `load(ParamServer&)` path, whose fitness_score_thresh default is 2.5."""
        p = dict(_DEFAULTS)
        if loaded:
            p["fitness_score_thresh"] = 2.5
        for k, v in (params or {}).items():
            if k in p:
                p[k] = type(_DEFAULTS[k])(v)
        self.__dict__.update(p)

    @staticmethod
    def weight(a: float, max_x: float, min_y: float, max_y: float, x: float) -> float:
        """hpp:41-44"""
        y = (1.0 - math.exp(-a * x)) / (1.0 - math.exp(-a * max_x))
        return min_y + (max_y - min_y) * y

    def from_fitness_score(self, fitness_score: float) -> np.ndarray:
        """.cpp:25-47 behind the calc_fitness_score call."""
        inf = np.eye(6)
        if self.use_const_inf_matrix:
            inf[:3, :3] /= self.const_stddev_x
            inf[3:, 3:] /= self.const_stddev_q
            return inf
        min_var_x, max_var_x = self.min_stddev_x ** 2, self.max_stddev_x ** 2
        min_var_q, max_var_q = self.min_stddev_q ** 2, self.max_stddev_q ** 2
        w_x = np.float32(self.weight(self.var_gain_a, self.fitness_score_thresh, min_var_x, max_var_x, fitness_score))   # `float w_x = ...`
        w_q = np.float32(self.weight(self.var_gain_a, self.fitness_score_thresh, min_var_q, max_var_q, fitness_score))
        inf[:3, :3] /= float(w_x)
        inf[3:, 3:] /= float(w_q)
        return inf

    def calc_information_matrix(self, reg, cloud1, cloud2, relpose) -> np.ndarray:
        """`reg`: a RegistrationHIP engine; cloud1 / cloud2: resident DeviceClouds (keyframe clouds).  The reference calls calc_fitness_score
        with its default max_range (DBL_MAX)."""
        if self.use_const_inf_matrix:
            return self.from_fitness_score(0.0)
        return self.from_fitness_score(reg.calc_fitness_score(cloud1, cloud2, relpose))
