"""Sampling-probability maps of run_type >= 2 (reference behaviour: prob_sampling_utils.py:50-94 ``gaussian_map``,
:146-165 ``combine_log_blend``; the EDT prior of car_env.py:100-101).  Host side, as in the reference: a 20 x 20
map evaluated once per plan / maze update; the per-candidate categorical draw stays in the host RNG order.
Bit-exact against the reference's outputs (tests/golden/geometry.npz, probmap_* keys): the floating-point
operations and their order are part of the contract, the code around them is not.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import distance_transform_edt


def edt_prior(maze_map):
    """Distance of every free cell to the nearest occupied one, normalised to a probability map."""
    dist = distance_transform_edt(1 - maze_map)
    return dist / np.sum(dist)


def _oriented_covariance(direction, spread_along, spread_across):
    normal = np.array([-direction[1], direction[0]])
    basis = np.stack([direction, normal], axis=1)
    return basis @ np.diag([spread_along ** 2, spread_across ** 2]) @ basis.T


def gaussian_map(robot, goal, size=(20, 20)):
    """Discrete 2-D Gaussian stretched along the robot -> goal line; ``robot`` / ``goal`` are (x, y) grid positions.
    The mean sits on the goal when it is near and slides towards the midpoint with distance; the spread grows with
    log(1 + distance); the robot's own cell gets no mass.  Returns ``(pdf, mean, covariance)``."""
    n_rows, n_cols = size
    (x_r, y_r), (x_g, y_g) = robot, goal
    step = np.array([x_g - x_r, y_g - y_r])
    length = np.sqrt(step[0] ** 2 + step[1] ** 2) + 1e-6
    heading = step / length if length > 1e-6 else np.array([1.0, 0.0])
    pull = -np.exp(-length / 15) + 1                       # 0 near the goal, -> 1 far away
    centre = (1 - pull) * np.array([x_g, y_g]) + pull * np.array([(x_r + x_g) / 2, (y_r + y_g) / 2])
    spread = 1.0 + 0.7 * np.log1p(length)
    cov = _oriented_covariance(heading, spread, 0.7 * spread)
    precision = np.linalg.inv(cov)
    rows, cols = np.mgrid[0:n_rows, 0:n_cols]
    offset = np.stack([cols, rows], axis=-1) - centre
    pdf = np.exp(-0.5 * np.sum((offset @ precision) * offset, axis=2))
    pdf[int(y_r), int(x_r)] = 0
    pdf /= pdf.sum()
    return pdf, centre, cov


def combine_log_blend(prior, gauss, beta=0.8, obstacle_mask=None, eps=1e-12):
    """Geometric blend prior^beta * gauss^(1 - beta), zero wherever the prior is zero (occupied cells), normalised.
    A vanishing blend falls back to the prior and then to a uniform map over the allowed cells."""
    allowed = np.ones(prior.shape, dtype=bool) if obstacle_mask is None else obstacle_mask
    candidates = (
        lambda: np.exp(beta * np.log(prior + eps) + (1.0 - beta) * np.log(gauss + eps)) * (prior > 0),
        lambda: prior,
        lambda: np.ones_like(prior, dtype=float),
    )
    for make in candidates:
        post = np.where(allowed, make(), 0.0)
        total = post.sum()
        if total > eps:
            break
    return post / total
