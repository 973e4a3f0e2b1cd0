"""Parameter container of the denoiser with the reference's state-dict layout.

``NoisePredNet`` stands where the reference's ``ConditionalUnet1DWithLocalMap``
(local_map_encoder.py:78-109) stands: an ``nn.Module`` whose ``state_dict()`` keys are
``encoder.resnet18.*`` / ``unet.*``, so ``load_state_dict(ckpt['noise_pred_net_state_dict'])``
works unchanged, and whose ``forward(sample, local_map, timestep, global_cond)`` runs the
HIP engine (MFMA implicit-GEMM kernels) instead of torch.nn layers.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .weights import noise_pred_net_param_shapes, pack_state_dict


class _Node(nn.Module):
    pass


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    m.register_parameter(parts[-1], param)


class NoisePredNet(nn.Module):
    def __init__(self, input_dim=2, embedding_dim=400, additional_global_cond_dim=7,
                 down_dims=(512, 1024, 2048), pred_horizon=64, local_map_size=20, seed=None, init=True):
        """``init=False``: parameters are allocated but not initialised -- for a net that is filled by ``load_state_dict`` right
        away (initialising 184 M parameters takes longer than loading them)."""
        super().__init__()
        self.input_dim, self.embedding_dim = input_dim, embedding_dim
        self.down_dims = tuple(int(d) for d in down_dims)
        self.global_cond_dim = additional_global_cond_dim
        self.pred_horizon, self.local_map_size = pred_horizon, local_map_size
        gen = torch.Generator().manual_seed(0 if seed is None else seed)
        for name, shape in noise_pred_net_param_shapes(input_dim, embedding_dim, additional_global_cond_dim,
                                                        down_dims).items():
            value = self._init(name, shape, gen) if init else torch.empty(shape)
            _attach(self, name, nn.Parameter(value, requires_grad=False))
        self._ctx = None
        self.precision = _lib.PREC_BF16
        self._reserved = 0

    @staticmethod
    def _init(name, shape, gen):
        """torch-default-like initialisation (not the reference's RNG stream: the reference ships
        no checkpoint; parity tests load oracle-seeded weights through load_state_dict)."""
        if len(shape) == 1:
            norm = (".block.1." in name) or (".bn" in name) or (".downsample.1." in name)
            if norm:
                return torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            return (torch.rand(shape, generator=gen) * 2 - 1) * 0.02
        fan_in = int(np.prod(shape[1:]))
        if "up_modules" in name and ".2.conv.weight" in name:
            fan_in = shape[1] * shape[2]
        if "encoder.resnet18" in name and len(shape) == 4:
            std = math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
            return torch.randn(shape, generator=gen) * std
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound

    # ------------------------------------------------------------------ engine binding
    def bind(self, ctx, precision=None, max_batch=None):
        """Upload the current parameters into ``ctx`` (repacked for MFMA inside the library)."""
        if precision is not None:
            self.precision = precision
        # in-memory hand-over: no checksum line (it guards blobs that were stored or shipped -- weights.pack_state_dict's
        # default -- and costs seconds on 0.7 GB in numpy)
        blob, manifest = pack_state_dict(self.state_dict(), pred_horizon=self.pred_horizon,
                                         local_map_size=self.local_map_size, checksum=False)
        ctx.load_weights(blob, manifest)
        ctx.weights_owner = self
        self._ctx = ctx
        self._bound_version = self.param_version()
        self._reserved = 0
        if max_batch:
            self.reserve(max_batch)
        return self

    def param_version(self):
        """Changes whenever a parameter is written in place (load_state_dict, copy_, add_ ...)."""
        return sum(int(p._version) for p in self.parameters())

    def is_current(self, ctx):
        """True when `ctx` holds THESE parameters (not another net's, not a stale copy)."""
        return self._ctx is ctx and ctx.weights_owner is self and getattr(self, "_bound_version", None) == self.param_version()

    def reserve(self, max_batch):
        if self._ctx is None:
            raise _lib.DitreeError("NoisePredNet.bind(ctx) first")
        if max_batch > self._reserved:
            self._ctx.denoise_reserve(max_batch, self.precision)
            self._reserved = max_batch

    def to(self, *args, **kwargs):           # parameters stay on the host; the engine owns the device copies
        return self

    def forward(self, sample, local_map, timestep, global_cond=None):
        """One velocity evaluation v = net(sample, local_map, t, cond) on the GPU
        (local_map_encoder.py:101-109).  ``timestep`` must be one value for the whole batch
        (the reference's sampler always passes ``ones * t``)."""
        if self._ctx is None:
            raise _lib.DitreeError("NoisePredNet is not bound to a device context: call .bind(ctx)")
        dev = self._ctx.device
        x = torch.as_tensor(sample, dtype=torch.float32, device=dev).contiguous()
        lm = torch.as_tensor(local_map, dtype=torch.float32, device=dev).contiguous()
        gc = torch.as_tensor(global_cond, dtype=torch.float32, device=dev).contiguous()
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).cpu()
        if not bool((t == t[0]).all()):
            raise ValueError("per-sample timesteps are not supported")
        if not self.is_current(self._ctx):
            self.bind(self._ctx)                  # another net was bound to the ctx, or the parameters changed
        self.reserve(x.shape[0])
        # one Euler step with dt = 1 from x: x1 = x + v  ->  v = x1 - x; t0 carries t / 20
        x1 = self._ctx.denoise(x, lm, gc, t0=np.array([float(t[0]) / 20.0], dtype=np.float32),
                               dt=np.array([1.0], dtype=np.float32), want_actions=False)
        return x1 - x
