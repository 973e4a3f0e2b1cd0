"""Deterministic, state-independent action tapes for parity runs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference planner, the oracle
planner and the HIP engine are all driven with the same (candidate, chunk) -> actions
tape, which stands in for the denoiser so that tree topology / flags / states can be
compared bit for bit (SURVEY.md section 7 "layered parity" (i) and (iii)).

A counter-based hash (splitmix64) makes the tape a pure function of
(seed, global candidate index, chunk index, step, action dim): any subset can be
generated in any order, vectorised, on any host.
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def _u01(key: np.ndarray) -> np.ndarray:
    return (_splitmix64(key) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class ActionTape:
    def __init__(self, seed: int, pred_horizon: int = 64):
        self.seed = np.uint64(seed)
        self.P = pred_horizon

    def _key(self, cand, chunk, t, d):
        with np.errstate(over="ignore"):
            k = self.seed * np.uint64(0x100000001B3)
            k = _splitmix64(k ^ cand.astype(np.uint64))
            k = _splitmix64(k ^ np.uint64(chunk * 1315423911 + 7))
            return k ^ (t.astype(np.uint64) * np.uint64(4) + d.astype(np.uint64) + np.uint64(1))

    def actions(self, cand_idx, chunk: int) -> np.ndarray:
        """(n,) global candidate indices -> (n, P, 2) float64 actions for that chunk."""
        cand = np.asarray(cand_idx, dtype=np.int64).reshape(-1, 1, 1)
        t = np.arange(self.P, dtype=np.int64).reshape(1, -1, 1)
        d = np.arange(2, dtype=np.int64).reshape(1, 1, 2)
        u = _u01(self._key(cand, chunk, t, d))                                  # per step / dim
        zero = np.zeros_like(t)
        base = _u01(self._key(cand, chunk, zero, d) ^ np.uint64(0xABCDEF))        # per chunk / dim
        style = _u01(self._key(cand, 977, zero, np.zeros_like(d)))[..., 0:1]     # per candidate
        steer_scale = np.where(style < 0.4, 0.05, np.where(style < 0.8, 0.6, 2.4))
        dD = -5.0 + 16.0 * base[..., 0:1] + 1.0 * (u[..., 0:1] - 0.5)            # may exceed +10: clipped by env
        dd = steer_scale * (2.0 * base[..., 1:2] - 1.0) + 0.05 * (u[..., 1:2] - 0.5)
        return np.concatenate([dD, dd], axis=2)

    def sampler(self):
        """OraclePlanner-compatible sampler callable."""
        def fn(cand_idx, chunk, state, prev_action, has_prev, cond_goal, local_map):
            return self.actions(cand_idx, chunk)
        return fn
