"""DDPM scheduler for the sampler's policy = 'diffusion' branch (reference: policies/fm_policy.py:164-182 driven by
``diffusers.schedulers.scheduling_ddpm.DDPMScheduler(num_train_timesteps=K, beta_schedule='squaredcos_cap_v2',
clip_sample=True, prediction_type='epsilon')``, run_scenarios.py:157-158).

``diffusers`` is a third-party dependency that is not part of the reference repository (and not installed here), so this file
follows the PUBLISHED algorithm -- Ho et al. 2020, with the cosine schedule of Nichol & Dhariwal 2021 as diffusers implements
it (float32 tensors, 'leading' timestep spacing, 'fixed_small' variance) -- and its parity with the reference's run-time
behaviour is UNPINNED.  Two uses:

* ``DDPMScheduler``: the surface the sampler touches (``set_timesteps``, ``timesteps``, ``step(...).prev_sample``), for hosts
  without diffusers;
* ``ddpm_tables(scheduler)``: the per-step constants of the on-device loop (include/ditree.h ditree_denoise_ddpm), read from
  THIS class or from a diffusers scheduler object of the same configuration.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


def _cosine_betas(n, max_beta=0.999):
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - abar((i + 1) / n) / abar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True,
                 prediction_type="epsilon", variance_type="fixed_small", clip_sample_range=1.0, **kw):
        if beta_schedule == "squaredcos_cap_v2":
            self.betas = _cosine_betas(num_train_timesteps)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}")
        if prediction_type != "epsilon" or variance_type != "fixed_small":
            raise NotImplementedError("covered: prediction_type 'epsilon', variance_type 'fixed_small' (the reference's configuration)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      prediction_type=prediction_type, variance_type=variance_type, clip_sample_range=clip_sample_range,
                                      timestep_spacing="leading", steps_offset=0, thresholding=False)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def previous_timestep(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def _coefficients(self, t):
        """float32 0-d tensors of one reverse step: sb, sa, c0, c1, variance."""
        t = int(t)
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c0 = (a_prev ** 0.5 * cur_b) / b_t
        c1 = cur_a ** 0.5 * b_prev / b_t
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        return b_t ** 0.5, a_t ** 0.5, c0, c1, var

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        sb, sa, c0, c1, var = (v.to(sample.device) for v in self._coefficients(timestep))
        x0 = (sample - sb * model_output) / sa
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        prev = c0 * x0 + c1 * sample
        if int(timestep) > 0:
            z = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + (var ** 0.5) * z
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


def ddpm_tables(scheduler, num_inference_steps):
    """-> (timesteps (K,) float32, coef (K, 5) float32 = sb, sa, c0, c1, sigma per step) of the on-device DDPM loop, or None
    when the scheduler is not the configuration the device step implements (the caller then keeps the host loop).  Works on this
    module's DDPMScheduler and on a diffusers DDPMScheduler (same attribute names)."""
    cfg = getattr(scheduler, "config", None)
    ac = getattr(scheduler, "alphas_cumprod", None)
    if cfg is None or ac is None:
        return None
    get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
    if (get("prediction_type", "epsilon") != "epsilon" or get("variance_type", "fixed_small") != "fixed_small" or not get("clip_sample", True)
            or float(get("clip_sample_range", 1.0)) != 1.0 or get("thresholding", False)):
        return None
    scheduler.set_timesteps(num_inference_steps)
    ts = [int(t) for t in scheduler.timesteps]
    n_train = int(get("num_train_timesteps"))
    ac = torch.as_tensor(ac, dtype=torch.float32).cpu()
    one = torch.tensor(1.0)
    rows = []
    for t in ts:
        prev_t = t - n_train // num_inference_steps
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        sigma = var ** 0.5 if t > 0 else torch.tensor(0.0)
        rows.append([float(b_t ** 0.5), float(a_t ** 0.5), float((a_prev ** 0.5 * cur_b) / b_t), float(cur_a ** 0.5 * b_prev / b_t), float(sigma)])
    return np.asarray(ts, dtype=np.float32), np.asarray(rows, dtype=np.float32)
