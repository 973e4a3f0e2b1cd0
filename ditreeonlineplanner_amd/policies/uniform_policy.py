"""Uniform random action sampler (reference import path policies/uniform_policy.py): one action drawn from the
environment's action space per call, shaped (1, action_dim).  Host only; kept so that the reference scripts' imports
resolve -- the batched engine is driven by DiffusionSampler."""
import numpy as np


class UniformSampler:
    def __init__(self, action_space):
        self.action_space = action_space

    def __call__(self, *args, **kwargs):
        draw = np.asarray(self.action_space.sample())
        return draw[np.newaxis, ...]
