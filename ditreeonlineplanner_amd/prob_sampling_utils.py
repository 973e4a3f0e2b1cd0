"""Sampling-probability maps of run_type >= 2 (reference: prob_sampling_utils.py:50-94 ``gaussian_map``,
:146-165 ``combine_log_blend``; the EDT prior of car_env.py:100-101).  Host side, as in the reference: a 20 x 20
map evaluated once per plan / maze update; the per-candidate categorical draw stays in the host RNG order."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import distance_transform_edt


def edt_prior(maze_map):
    """Distance to the nearest occupied cell, as a probability map (car_env.py:100-101,120-121)."""
    prior = distance_transform_edt(1 - maze_map)
    return prior / np.sum(prior)


def gaussian_map(robot, goal, size=(20, 20)):
    """Discrete 2-D Gaussian elongated along robot -> goal; ``robot`` / ``goal`` are (x, y) grid positions.
    Returns ``(pdf, mean, Sigma)`` like the reference."""
    H, W = size
    rx, ry = robot
    gx, gy = goal
    dx, dy = gx - rx, gy - ry
    d = np.sqrt(dx ** 2 + dy ** 2) + 1e-6
    u = np.array([dx, dy]) / d if d > 1e-6 else np.array([1.0, 0.0])       # unit vector robot -> goal
    v = np.array([-u[1], u[0]])
    # mean: the goal when close, sliding to the midpoint with distance
    w = -np.exp(-d / 15) + 1
    mean = (1 - w) * np.array([gx, gy]) + w * np.array([(rx + gx) / 2, (ry + gy) / 2])
    sigma_long = 1.0 + 0.7 * np.log1p(d)
    sigma_side = 0.7 * sigma_long
    R = np.stack([u, v], axis=1)
    Sigma = R @ np.diag([sigma_long ** 2, sigma_side ** 2]) @ R.T
    Sigma_inv = np.linalg.inv(Sigma)
    ys, xs = np.mgrid[0:H, 0:W]
    diff = np.stack([xs, ys], axis=-1) - mean
    pdf = np.exp(-0.5 * np.sum((diff @ Sigma_inv) * diff, axis=2))            # Mahalanobis form
    pdf[int(ry), int(rx)] = 0
    pdf /= pdf.sum()
    return pdf, mean, Sigma


def combine_log_blend(prior, gauss, beta=0.8, obstacle_mask=None, eps=1e-12):
    """posterior ~ prior^beta * gauss^(1 - beta), zero wherever the prior is zero (obstacles)."""
    post = np.exp(beta * np.log(prior + eps) + (1.0 - beta) * np.log(gauss + eps)) * (prior > 0)
    free = obstacle_mask if obstacle_mask is not None else np.ones_like(post, dtype=bool)
    post = np.where(free, post, 0.0)
    s = post.sum()
    if s <= eps:                                         # degenerate: the prior alone, then uniform over free cells
        post = np.where(free, prior, 0.0)
        s = post.sum()
        if s <= eps:
            post = np.where(free, 1.0, 0.0)
            s = post.sum()
    return post / s
