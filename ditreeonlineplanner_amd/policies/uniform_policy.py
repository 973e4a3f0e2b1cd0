"""Uniform random action sampler (reference: policies/uniform_policy.py:3-8).

``UniformSampler(action_space)()`` returns one action drawn from the environment's action space, shaped
(1, action_dim), as the reference's class does.  The reference never wires it into a planner (its only uses are
commented out, run_scenarios.py:275-283, and a non-``nn.Module`` sampler is not even stored, base_planner.py:56-57), so
what a planner does with it is the build's definition (BASELINE config 1 "plumbing"): ``RRT_Planner``
asks a sampler without ``ensure_bound`` for the action sequences of a whole round and by-passes the denoiser
(``ditree_round_params.inject_actions``).  ``sample_round`` is that protocol: uniform actions, one per env step, a pure
function of (seed, global candidate index) so that the tree does not depend on the round size or the rank count."""
import numpy as np


class UniformSampler:
    def __init__(self, action_space, seed=None):
        self.action_space = action_space
        self.seed = seed

    def __call__(self, *args, **kwargs):
        draw = np.asarray(self.action_space.sample())
        return draw[np.newaxis, ...]

    def sample_round(self, first_candidate, B, n_chunks, pred_horizon):
        """-> (B, n_chunks, pred_horizon, action_dim) float64, uniform in [action_space.low, action_space.high]."""
        lo = np.asarray(self.action_space.low, dtype=np.float64)
        hi = np.asarray(self.action_space.high, dtype=np.float64)
        out = np.empty((B, n_chunks, pred_horizon, lo.size))
        base = 0 if self.seed is None else int(self.seed)
        for b in range(B):
            g = np.random.default_rng([base, first_candidate + b])
            out[b] = g.uniform(lo, hi, size=(n_chunks, pred_horizon, lo.size))
        return out
