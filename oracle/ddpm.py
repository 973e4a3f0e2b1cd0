"""numpy restatement of the DDPM reverse process the reference's sampler runs for policy = 'diffusion'
(policies/fm_policy.py:164-182 with diffusers' DDPMScheduler configured at run_scenarios.py:157-158) -- TEST INFRASTRUCTURE ONLY.

``diffusers`` is third party and absent from both the reference repository and this image: this follows the PUBLISHED algorithm
(Ho et al. 2020, eq. 7 / 15 with x0 clipping; cosine schedule of Nichol & Dhariwal 2021, betas capped at 0.999; float32 like the
scheduler's tensors) and is written independently of the product's ditreeonlineplanner_amd/ddpm.py.  **Parity unpinned.**"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def cosine_alphas_cumprod(n):
    t = np.arange(n + 1, dtype=np.float64) / n
    abar = np.cos((t + 0.008) / 1.008 * np.pi / 2) ** 2
    betas = np.minimum(1 - abar[1:] / abar[:-1], 0.999).astype(f32)
    return np.cumprod((f32(1) - betas).astype(f32), dtype=f32)


def reverse_process(eps_fn, x, n_train, step_noise):
    """K = n_train reverse steps (one per training timestep, as the reference sets num_train_timesteps = planning iterations):
    ``eps_fn(x (B, P, D) f32, t) -> eps`` is the network, ``step_noise[k]`` the standard-normal z of step k (unused at t = 0)."""
    ac = cosine_alphas_cumprod(n_train)
    x = x.astype(f32)
    for k, t in enumerate(range(n_train - 1, -1, -1)):
        eps = eps_fn(x, float(t)).astype(f32)
        a_t = ac[t]
        a_prev = ac[t - 1] if t > 0 else f32(1)
        x0 = np.clip((x - np.sqrt(f32(1) - a_t) * eps) / np.sqrt(a_t), f32(-1), f32(1))
        cur_a = a_t / a_prev
        mean = (np.sqrt(a_prev) * (f32(1) - cur_a) / (f32(1) - a_t)) * x0 + (np.sqrt(cur_a) * (f32(1) - a_prev) / (f32(1) - a_t)) * x
        if t > 0:
            var = max((f32(1) - a_prev) / (f32(1) - a_t) * (f32(1) - cur_a), f32(1e-20))
            mean = mean + np.sqrt(f32(var)) * step_noise[k].astype(f32)
        x = mean.astype(f32)
    return x
