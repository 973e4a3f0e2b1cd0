"""CPU restatement (numpy f64) of the MPPI controller step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The controller `run_scenarios_with_lidar_MPPI.py:10,339-449` imports (`MPPI.mppi.MPPI`) is NOT part of the reference
repository: there is no reference algorithm, golden vector or fixture for it.  **Parity unpinned.**  What
this module restates is the build's own definition (include/ditree.h `ditree_mppi_step`, DESIGN.md "MPPI"), written
independently of the HIP kernels (vectorised over the rollouts, different summation structure) on top of the pinned pieces
of the oracle: the car dynamics (car_env.py:356-396), the two-ball collision test (common/map_utils.py:103-115) and the goal
radius (car_env.py:341-354).
"""
from __future__ import annotations

import numpy as np

from . import geometry as G

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def device_noise(seed, counter, K, T, sigma):
    """The on-device generator of ditree_mppi_step (noise == NULL): eps (K, T, 2), a pure function of (seed, counter, k, t)."""
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed) ^ _splitmix64(np.uint64(counter)))
        k = np.arange(K, dtype=np.uint64).reshape(-1, 1)
        t = np.arange(T, dtype=np.uint64).reshape(1, -1)
        h = _splitmix64(h ^ ((k * np.uint64(0xD1B54A32D192ED03)) & _M))
        h = _splitmix64(h ^ t)
        h2 = _splitmix64(h)
    u1 = ((h >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
    u2 = (h2 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    r = np.sqrt(-2.0 * np.log(u1))
    a = 6.283185307179586 * u2
    return np.stack([sigma[0] * (r * np.cos(a)), sigma[1] * (r * np.sin(a))], axis=-1)


def rollout_costs(maze, state, U, path_xy, goal_xy, noise, lam, sigma, w_track, w_progress, w_collision, w_goal,
                  window_back, window_fwd):
    """-> (costs (K,), flags (K,) 0 / 1 goal / 2 collided, i0).  noise (K, T, 2); rollout 0 runs without noise."""
    state = np.asarray(state, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    path = np.asarray(path_xy, dtype=np.float64)
    eps = np.array(noise, dtype=np.float64)
    eps[0] = 0.0
    K, T = eps.shape[:2]
    P = len(path)
    d0 = (path[:, 0] - state[0]) ** 2 + (path[:, 1] - state[1]) ** 2
    i0 = int(np.argmin(d0))
    x = np.tile(state, (K, 1))
    cost = np.zeros(K)
    ip = np.full(K, i0, dtype=np.int64)
    flags = np.zeros(K, dtype=np.int32)
    alive = np.ones(K, dtype=bool)
    for t in range(T):
        idx = np.nonzero(alive)[0]
        if idx.size == 0:
            break
        xn = G.car_step(x[idx], U[t] + eps[idx, t])
        x[idx] = xn
        coll = G.is_colliding_car(xn, maze)
        reached = G.goal_reached(xn, goal_xy)
        # windowed nearest path point, first occurrence of the minimum: all alive rollouts at once over the (window_back +
        # window_fwd + 1)-wide index window, positions outside [max(ip - back, 0), min(ip + fwd, P - 1)] masked out
        w = ip[idx][:, None] + np.arange(-window_back, window_fwd + 1)[None, :]
        ok = (w >= 0) & (w <= P - 1)
        wc = np.clip(w, 0, P - 1)
        dd = (path[wc, 0] - xn[:, None, 0]) ** 2 + (path[wc, 1] - xn[:, None, 1]) ** 2
        dd = np.where(ok, dd, np.inf)
        j = np.argmin(dd, axis=1)
        ip[idx] = w[np.arange(idx.size), j]
        d2 = dd[np.arange(idx.size), j]
        cost[idx] = cost[idx] + w_track * d2
        cost[idx] = cost[idx] + lam * ((U[t, 0] * eps[idx, t, 0]) / (sigma[0] * sigma[0]) + (U[t, 1] * eps[idx, t, 1]) / (sigma[1] * sigma[1]))
        cost[idx[coll]] = cost[idx[coll]] + w_collision
        flags[idx[coll]] = 2
        only_goal = reached & ~coll
        cost[idx[only_goal]] = cost[idx[only_goal]] - w_goal
        flags[idx[only_goal]] = 1
        alive[idx[coll | reached]] = False
    cost = cost + w_progress * (P - 1 - ip).astype(np.float64)
    return cost, flags, i0


def update(U, costs, noise, lam):
    """-> (U_new, normalised weights, beta, eta, effective sample size)."""
    eps = np.array(noise, dtype=np.float64)
    eps[0] = 0.0
    beta = float(np.min(costs))
    w = np.exp(-(costs - beta) / lam)
    eta = float(np.sum(w))
    dU = np.tensordot(w, eps, axes=(0, 0)) / eta
    return np.asarray(U, dtype=np.float64) + dU, w / eta, beta, eta, eta * eta / float(np.sum(w * w))


def execute(maze, state, U, goal_xy):
    """-> (state', action (2,), status 0 / 1 goal / 2 collided, U shifted)."""
    U = np.asarray(U, dtype=np.float64)
    a = np.clip(U[0], G.ACT_LOW, G.ACT_HIGH)
    x = G.car_step(np.asarray(state, dtype=np.float64)[None], a[None])[0]
    if bool(G.is_colliding_car(x[None], maze)[0]):
        return np.asarray(state, dtype=np.float64).copy(), a, 2, np.zeros_like(U)
    status = 1 if bool(G.goal_reached(x[None], goal_xy)[0]) else 0
    Un = U.copy()
    Un[:-1] = U[1:]
    return x, a, status, Un


# --------------------------------------------------------------------------- the same controller on the ant slot (config 5 as written)
# 29-d state, 8-d controls, the build's stand-in crawler model (oracle/ant.py ant_model_step; NOT MuJoCo), the reference's ant
# collision test and goal radius.  Neither the controller nor the dynamics exist in the reference: parity unpinned.
def device_noise_ant(seed, counter, k0, K, T, sigma):
    """ditree_mppi_step_ant's on-device generator: eps (K, T, 8); pair p of (seed, counter, GLOBAL k, t) gives dims 2p, 2p + 1."""
    sigma = np.asarray(sigma, dtype=np.float64)
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed) ^ _splitmix64(np.uint64(counter)))
        k = (np.arange(K, dtype=np.uint64) + np.uint64(k0)).reshape(-1, 1)
        t = np.arange(T, dtype=np.uint64).reshape(1, -1)
        h = _splitmix64(h ^ ((k * np.uint64(0xD1B54A32D192ED03)) & _M))
        h = _splitmix64(h ^ t)
        out = np.zeros((K, T, 8))
        for p in range(4):
            h1 = _splitmix64(h ^ ((np.uint64(p + 1) * np.uint64(0xA24BAED4963EE407)) & _M))
            h2 = _splitmix64(h1)
            u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
            u2 = (h2 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
            r = np.sqrt(-2.0 * np.log(u1))
            a = 6.283185307179586 * u2
            out[..., 2 * p] = sigma[2 * p] * (r * np.cos(a))
            out[..., 2 * p + 1] = sigma[2 * p + 1] * (r * np.sin(a))
    return out


def rollout_costs_ant(maze, state, U, path_xy, desired_xy, noise, lam, sigma, w_track, w_progress, w_collision, w_goal, window_back,
                      window_fwd, s_global=4.0, ball_radius=1.2, goal_radius=None, k0=0):
    """-> (costs (K,), flags (K,), i0); noise (K, T, 8); GLOBAL rollout 0 (k0 + k == 0) runs without noise."""
    from . import ant as OA
    state = np.asarray(state, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    path = np.asarray(path_xy, dtype=np.float64)
    eps = np.array(noise, dtype=np.float64)
    if k0 == 0:
        eps[0] = 0.0
    sigma = np.asarray(sigma, dtype=np.float64)
    goal_radius = OA.ANT_GOAL_FACTOR * s_global if goal_radius is None else goal_radius
    K, T = eps.shape[:2]
    P = len(path)
    d0 = (path[:, 0] - state[0]) ** 2 + (path[:, 1] - state[1]) ** 2
    i0 = int(np.argmin(d0))
    x = np.tile(state, (K, 1))
    cost = np.zeros(K)
    ip = np.full(K, i0, dtype=np.int64)
    flags = np.zeros(K, dtype=np.int32)
    alive = np.ones(K, dtype=bool)
    for t in range(T):
        idx = np.nonzero(alive)[0]
        if idx.size == 0:
            break
        xn = OA.ant_model_step(x[idx], U[t] + eps[idx, t])
        x[idx] = xn
        coll = OA.is_colliding_ant(xn, maze, ball_radius, s_global)
        d = xn[:, :2] - np.asarray(desired_xy, dtype=np.float64)
        reached = G.norm2(d[:, 0], d[:, 1]) < goal_radius
        w = ip[idx][:, None] + np.arange(-window_back, window_fwd + 1)[None, :]
        ok = (w >= 0) & (w <= P - 1)
        wc = np.clip(w, 0, P - 1)
        dd = (path[wc, 0] - xn[:, None, 0]) ** 2 + (path[wc, 1] - xn[:, None, 1]) ** 2
        dd = np.where(ok, dd, np.inf)
        j = np.argmin(dd, axis=1)
        ip[idx] = w[np.arange(idx.size), j]
        cost[idx] = cost[idx] + w_track * dd[np.arange(idx.size), j]
        ctrl = np.zeros(idx.size)
        for dim in range(8):
            ctrl = ctrl + (U[t, dim] * eps[idx, t, dim]) / (sigma[dim] * sigma[dim])
        cost[idx] = cost[idx] + lam * ctrl
        cost[idx[coll]] = cost[idx[coll]] + w_collision
        flags[idx[coll]] = 2
        only_goal = reached & ~coll
        cost[idx[only_goal]] = cost[idx[only_goal]] - w_goal
        flags[idx[only_goal]] = 1
        alive[idx[coll | reached]] = False
    cost = cost + w_progress * (P - 1 - ip).astype(np.float64)
    return cost, flags, i0


def update_ant(U, costs, noise, lam, k0=0):
    eps = np.array(noise, dtype=np.float64)
    if k0 == 0:
        eps[0] = 0.0
    beta = float(np.min(costs))
    w = np.exp(-(costs - beta) / lam)
    eta = float(np.sum(w))
    return np.asarray(U, dtype=np.float64) + np.tensordot(w, eps, axes=(0, 0)) / eta, w / eta, beta, eta, eta * eta / float(np.sum(w * w))


def execute_ant(maze, state, U, desired_xy, s_global=4.0, ball_radius=1.2):
    from . import ant as OA
    U = np.asarray(U, dtype=np.float64)
    a = np.clip(U[0], -1.0, 1.0)
    x = OA.ant_model_step(np.asarray(state, dtype=np.float64)[None], a[None])[0]
    if bool(OA.is_colliding_ant(x[None], maze, ball_radius, s_global)[0]):
        return np.asarray(state, dtype=np.float64).copy(), a, 2, np.zeros_like(U)
    status = 1 if bool(OA.ant_goal_reached(x[None], desired_xy, s_global)[0]) else 0
    Un = U.copy()
    Un[:-1] = U[1:]
    return x, a, status, Un
