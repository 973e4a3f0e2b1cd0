"""DiffusionSampler facade (reference: policies/fm_policy.py:10-212) over the HIP engine.

Same constructor and ``forward(obs_seq, prev_actions, goal, local_map) -> ndarray (B, pred_horizon,
action_dim) float64``.  Conditioning vector, local-map scaling, K flow steps and the action
un-normalisation all run on the GPU (ditree_cond_vector / ditree_denoise); only the tiny
observation arrays cross PCIe.  Car configuration (``env_id`` containing "car",
``policy='flow_matching'``, ``prediction_type='actions'``) -- the other branches of the reference
class are outside this round's scope and raise NotImplementedError.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
from torch import nn

from .. import _lib
from ..common.fm_utils import get_timesteps
from ..ops import default_context

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def load_metadata(env_id: str):
    """``metadata/{env_id}.pt`` relative to the CWD (fm_policy.py:28-32) when it can be read with the
    safe loader, else the packaged JSON copy of the same numbers; FileNotFoundError otherwise."""
    path = f"metadata/{env_id}.pt"
    if os.path.exists(path):
        try:
            import numpy.core.multiarray as _m
            with torch.serialization.safe_globals([_m._reconstruct, np.ndarray, np.dtype, type(np.dtype("f8"))]):
                md = torch.load(path, weights_only=True)
            return {k: np.asarray(v, dtype=np.float64) for k, v in md.items()}
        except Exception:
            pass
    js = os.path.join(_DATA, f"metadata_{env_id}.json")
    if os.path.exists(js):
        with open(js) as f:
            return {k: np.asarray(v, dtype=np.float64) for k, v in json.load(f).items() if not k.startswith("_")}
    raise FileNotFoundError(f"Metadata not found at {path}")


class DiffusionSampler(nn.Module):
    def __init__(self, noise_pred_net, noise_scheduler, env_id, policy, pred_horizon, action_dim,
                 prediction_type="actions", obs_history=1, action_history=1, num_diffusion_iters=100,
                 position_conditioned=False, goal_conditioned=True, local_map_conditioned=True, local_map_size=16,
                 metadata=None, ctx=None, precision=None):
        super().__init__()
        self.is_ant = "ant" in env_id.lower()
        if not ("car" in env_id.lower() or self.is_ant) or policy not in ("flow_matching", "diffusion") or prediction_type != "actions":
            raise NotImplementedError("covered: carmaze and antmaze, flow_matching / diffusion, action prediction")
        if policy == "diffusion" and noise_scheduler is None:
            raise ValueError("policy='diffusion' needs a noise scheduler (set_timesteps / timesteps / step)")
        if action_history != 1 or position_conditioned or not goal_conditioned or obs_history != (3 if self.is_ant else 1):
            raise NotImplementedError("covered: car (obs_history 1) and ant (obs_history 3, run_scenarios.py:123-132), "
                                      "action_history 1, goal conditioned")
        if action_dim != (8 if self.is_ant else 2):
            raise NotImplementedError("action_dim: 2 (car) or 8 (ant)")
        self.metadata = metadata if metadata is not None else load_metadata(env_id)
        self.env_id, self.policy, self.prediction_type = env_id, policy, prediction_type
        self.action_dim, self.pred_horizon = action_dim, pred_horizon
        self.obs_history, self.action_history = obs_history, action_history
        self.num_diffusion_iters = num_diffusion_iters
        self.local_map_size = local_map_size
        self.noise_pred_net = noise_pred_net
        self.noise_scheduler = noise_scheduler
        self._ctx = ctx
        # Default: the f32-class instantiation (f16 hi + lo planes, 3 MFMAs per product), which reproduces the fp32
        # reference to 1e-6; plain bf16 / f16 (3x the rate, 8 / 11 significand bits) are explicit opt-ins.  Denoiser sizes
        # whose channels are not multiples of 256 have no split tiles: those default to the f32 MFMA instantiation.
        if precision is None:
            dims = getattr(noise_pred_net, "down_dims", ())
            precision = _lib.PREC_F16X3 if dims and all(d % 256 == 0 for d in dims) and pred_horizon % 16 == 0 else _lib.PREC_F32
        self.precision = precision
        t0, dt = get_timesteps("exp", num_diffusion_iters, exp_scale=4.0)
        self.t0, self.dt = t0.numpy().copy(), dt.numpy().copy()
        self.norm = np.concatenate([self.metadata["Observations_mean"], self.metadata["Observations_std"],
                                    self.metadata["Actions_mean"], self.metadata["Actions_std"]]).astype(np.float64)
        self.device = "cuda"

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def to(self, *a, **k):
        return self

    def ddpm_tables(self):
        """(timesteps, coef) of the on-device DDPM loop when ``noise_scheduler`` is a DDPM scheduler of the configuration the
        device step implements (epsilon prediction, clip_sample, fixed_small variance: run_scenarios.py:157-158), else None."""
        if self.policy != "diffusion":
            return None
        if getattr(self, "_ddpm_tables", None) is None:
            from ..ddpm import ddpm_tables
            self._ddpm_tables = (ddpm_tables(self.noise_scheduler, self.num_diffusion_iters),)
        return self._ddpm_tables[0]

    def ensure_bound(self, max_batch):
        net = self.noise_pred_net
        # what the tensor shapes do not tell the library: sequence length and map size come from the sampler
        if net.pred_horizon != self.pred_horizon or net.local_map_size != self.local_map_size:
            net.pred_horizon, net.local_map_size = self.pred_horizon, self.local_map_size
            net._ctx = None
        if not net.is_current(self.ctx) or net.precision != self.precision:
            net.bind(self.ctx, precision=self.precision)
        net.reserve(max_batch)

    def forward(self, obs_seq, prev_actions, goal=None, local_map=None):
        ctx = self.ctx
        dev = ctx.device
        obs_seq = np.asarray(obs_seq, dtype=np.float64)
        if obs_seq.ndim == 1:
            obs_seq = obs_seq[None]
        if obs_seq.ndim == 2:
            obs_seq = obs_seq[:, None]
        B = obs_seq.shape[0]
        if self.is_ant:
            return self._forward_ant(obs_seq, prev_actions, goal, local_map)
        state = torch.as_tensor(np.ascontiguousarray(obs_seq[:, -1, :]), device=dev)           # obs_history = 1
        if prev_actions is not None:
            pa = np.asarray(prev_actions, dtype=np.float64)
            if pa.ndim == 2:
                pa = pa[None]
            if pa.shape[1] == 0:                                  # fm_policy.py:116-121 with an empty history
                last = np.zeros((B, self.action_dim))
            else:
                last = np.broadcast_to(pa[:, -1, :], (B, self.action_dim))
            has_prev = np.ones(B, dtype=np.uint8)
        else:
            last = np.zeros((B, self.action_dim))
            has_prev = np.zeros(B, dtype=np.uint8)
        g = np.broadcast_to(np.asarray(goal, dtype=np.float64).reshape(-1, 2), (B, 2))
        cond = ctx.cond_vector(state, torch.as_tensor(np.ascontiguousarray(last, dtype=np.float64), device=dev),
                               torch.as_tensor(has_prev, device=dev),
                               torch.as_tensor(np.ascontiguousarray(g, dtype=np.float64), device=dev), self.local_map_size, self.norm)
        lm = torch.as_tensor(local_map, dtype=torch.float32, device=dev)
        if lm.dim() == 2:
            lm = lm.unsqueeze(0)
        lm = (lm * 2 - 1).contiguous()                             # fm_policy.py:152
        noise = torch.randn((B, self.pred_horizon, self.action_dim), device=dev)     # :158-159
        self.ensure_bound(B)
        if self.policy == "diffusion":
            tables = self.ddpm_tables()
            if tables is not None:
                # fm_policy.py:164-182 with a DDPM scheduler of the reference's configuration: the whole reverse process on the
                # device (ditree_denoise_ddpm); the step noise is drawn here in the order scheduler.step would draw it
                K = len(tables[0])
                z = torch.randn((K, B, self.pred_horizon, self.action_dim), device=dev).permute(1, 0, 2, 3).contiguous()
                return ctx.denoise_ddpm(noise, z, lm, cond, tables[0], tables[1], act_norm=self.norm[12:16]).cpu().numpy()
            # any other scheduler object drives the loop itself (fm_policy.py:164-182); every
            # model call is one raw network evaluation on the device, the map embedding is computed once
            self.noise_scheduler.set_timesteps(self.num_diffusion_iters)
            naction = noise
            for i, k in enumerate(self.noise_scheduler.timesteps):
                noise_pred = ctx.denoise_eval(naction.contiguous(), lm, cond, float(k), reuse_encoder=i > 0)
                naction = self.noise_scheduler.step(model_output=noise_pred, timestep=k, sample=naction).prev_sample
                naction = naction.to(torch.float32)
            x = naction.detach().to("cpu").numpy()
            return x * self.metadata["Actions_std"] + self.metadata["Actions_mean"]          # :201-203
        actions = ctx.denoise(noise, lm, cond, t0=self.t0, dt=self.dt, act_norm=self.norm[12:16], want_actions=True)
        return actions.cpu().numpy()


def _forward_ant(self, obs_seq, prev_actions, goal, local_map):
    """policies/fm_policy.py:53-212, antmaze branch: (B, h, 29) observations -> (B, 16, 8) float64 actions."""
    ctx = self.ctx
    dev = ctx.device
    B = obs_seq.shape[0]
    if obs_seq.shape[2] != 29:
        raise ValueError("antmaze observations are (B, h, 29): x, y + the 27 MuJoCo observations")
    hist = np.ascontiguousarray(obs_seq[:, -3:, :])                                          # :96-102
    if prev_actions is not None:
        pa = np.asarray(prev_actions, dtype=np.float64)
        if pa.ndim == 2:
            pa = pa[None]
        last = np.zeros((B, 8)) if pa.shape[1] == 0 else np.broadcast_to(pa[:, -1, :], (B, 8))
        has_prev = np.ones(B, dtype=np.uint8)
    else:
        last, has_prev = np.zeros((B, 8)), np.zeros(B, dtype=np.uint8)
    g = np.broadcast_to(np.asarray(goal, dtype=np.float64).reshape(-1, 2), (B, 2))
    norm = np.concatenate([self.metadata["Observations_mean"], self.metadata["Observations_std"],
                           self.metadata["Actions_mean"], self.metadata["Actions_std"]]).astype(np.float64)
    cond = ctx.cond_vector_ant(torch.as_tensor(hist, device=dev), torch.as_tensor(np.ascontiguousarray(last), device=dev),
                               torch.as_tensor(has_prev, device=dev),
                               torch.as_tensor(np.ascontiguousarray(g, dtype=np.float64), device=dev), self.local_map_size, norm)
    lm = torch.as_tensor(local_map, dtype=torch.float32, device=dev)
    if lm.dim() == 2:
        lm = lm.unsqueeze(0)
    lm = (lm * 2 - 1).contiguous()
    noise = torch.randn((B, self.pred_horizon, self.action_dim), device=dev)
    self.ensure_bound(B)
    if self.policy == "diffusion":
        tables = self.ddpm_tables()
        if tables is not None:
            K = len(tables[0])
            z = torch.randn((K, B, self.pred_horizon, self.action_dim), device=dev).permute(1, 0, 2, 3).contiguous()
            return ctx.denoise_ddpm(noise, z, lm, cond, tables[0], tables[1],
                                    act_norm=np.concatenate([self.metadata["Actions_mean"], self.metadata["Actions_std"]])).cpu().numpy()
        self.noise_scheduler.set_timesteps(self.num_diffusion_iters)
        naction = noise
        for i, k in enumerate(self.noise_scheduler.timesteps):
            noise_pred = ctx.denoise_eval(naction.contiguous(), lm, cond, float(k), reuse_encoder=i > 0)
            naction = self.noise_scheduler.step(model_output=noise_pred, timestep=k, sample=naction).prev_sample.to(torch.float32)
        return naction.detach().to("cpu").numpy() * self.metadata["Actions_std"] + self.metadata["Actions_mean"]
    act_norm = np.concatenate([self.metadata["Actions_mean"], self.metadata["Actions_std"]])
    return ctx.denoise(noise, lm, cond, t0=self.t0, dt=self.dt, act_norm=act_norm, want_actions=True).cpu().numpy()


DiffusionSampler._forward_ant = _forward_ant
