"""TEST INFRASTRUCTURE -- CPU restatement of the sampling-probability maps of run_type >= 2
(reference: prob_sampling_utils.py:50-94 gaussian_map, :146-165 combine_log_blend; car_env.py:100-137 the
EDT prior and the wiring per run_type; planners/base_planner.py:157-160,181-184 the categorical cell draw).
Pinned by tests/golden/geometry.npz (probmap_* keys: outputs of the reference's own functions)."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import distance_transform_edt


def edt_prior(maze):
    """car_env.py:100-101,120-121: Euclidean distance to the nearest occupied cell, normalised to sum 1."""
    d = distance_transform_edt(1 - np.asarray(maze))
    return d / np.sum(d)


def gaussian_map(robot, goal, size=(20, 20)):
    """prob_sampling_utils.py:50-94.  robot / goal are (x, y) grid positions; returns (pdf, mean, Sigma).
    An anisotropic Gaussian stretched along robot -> goal whose mean slides from the goal (near) to the
    midpoint (far); the robot's own cell gets zero mass."""
    n_rows, n_cols = size
    rx, ry = robot
    gx, gy = goal
    dx, dy = gx - rx, gy - ry
    dist = np.sqrt(dx ** 2 + dy ** 2) + 1e-6
    along = np.array([dx, dy]) / dist if dist > 1e-6 else np.array([1.0, 0.0])
    across = np.array([-along[1], along[0]])
    mid = np.array([(rx + gx) / 2, (ry + gy) / 2])
    w = -np.exp(-dist / 15) + 1
    mean = (1 - w) * np.array([gx, gy]) + w * mid
    s_long = 1.0 + 0.7 * np.log1p(dist)
    s_side = 0.7 * s_long
    rot = np.stack([along, across], axis=1)
    cov = rot @ np.diag([s_long ** 2, s_side ** 2]) @ rot.T
    prec = np.linalg.inv(cov)
    yy, xx = np.mgrid[0:n_rows, 0:n_cols]
    delta = np.stack([xx, yy], axis=-1) - mean
    maha = np.sum((delta @ prec) * delta, axis=2)
    pdf = np.exp(-0.5 * maha)
    pdf[int(ry), int(rx)] = 0
    pdf /= pdf.sum()
    return pdf, mean, cov


def combine_log_blend(prior, gauss, beta=0.8, eps=1e-12):
    """prob_sampling_utils.py:146-165 without an obstacle mask (the env never passes one): geometric blend
    prior^beta * gauss^(1-beta), zero where the prior is zero, normalised; degenerate sums fall back to the
    prior, then to uniform."""
    blend = np.exp(beta * np.log(prior + eps) + (1.0 - beta) * np.log(gauss + eps)) * (prior > 0)
    total = blend.sum()
    if total <= eps:
        blend = prior.copy()
        total = blend.sum()
        if total <= eps:
            blend = np.where(np.ones_like(blend, dtype=bool), 1.0, 0.0)
            total = blend.sum()
    return blend / total


def draw_cell(rng, prob_map):
    """base_planner.py:157-160: one categorical draw over the flattened map -> (row, col)."""
    flat = rng.choice(prob_map.size, size=1, p=prob_map.ravel())
    row, col = np.unravel_index(flat, prob_map.shape)
    return int(row[0]), int(col[0])
