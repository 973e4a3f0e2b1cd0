"""Restatement of the flow-matching sampler's pre/post-processing (car config).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference sites: policies/fm_policy.py:53-212 (DiffusionSampler.forward),
common/fm_utils.py:4-17 (get_timesteps).
"""
from __future__ import annotations

import numpy as np
import torch

# metadata/carmaze.pt values.  The file is a pickle that torch.load(weights_only=True)
# refuses (numpy globals); the numbers below were read from its raw bytes with
# pickletools (nothing executed) and agree with SURVEY.md section 8(a) a6.
CAR_META = {
    "Observations_mean": np.array([0.0, 0.0, 0.0, 5.0, 0.5, 0.0]),
    "Observations_std": np.array([5.0, 5.0, 3.141592653589793, 5.0, 0.5, 0.4]),
    "Actions_mean": np.array([0.45102226669605805, 0.0]),
    "Actions_std": np.array([1.0061299587120194, 0.9234329966426903]),
}


def _load_ant_meta():
    """metadata/antmaze.pt of the reference: raw buffers read with pickletools (tests/golden/extract_metadata.py writes
    ditreeonlineplanner_amd/data/metadata_antmaze.json; nothing is unpickled)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ditreeonlineplanner_amd", "data",
                        "metadata_antmaze.json")
    with open(path) as f:
        d = json.load(f)
    return {k: np.asarray(v, dtype=np.float64) for k, v in d.items() if not k.startswith("_")}


ANT_META = _load_ant_meta()


def q_to_rot6d(q):
    """common/se3_utils.py:177-189 (q = [x, y, z, w])."""
    qw, qx, qy, qz = q[..., 3], q[..., 0], q[..., 1], q[..., 2]
    return np.stack([1 - 2 * (qy ** 2 + qz ** 2), 2 * (qx * qy + qw * qz), 2 * (qx * qz - qw * qy),
                     2 * (qx * qy - qw * qz), 1 - 2 * (qx ** 2 + qz ** 2), 2 * (qy * qz + qw * qx)], axis=-1)


def ant_cond_vector(obs_seq, prev_action, has_prev, goal_xy, local_map_size=16, obs_history=3, meta=None):
    """fm_policy.py:60-143 for antmaze (obs_history 3, action_history 1, position_conditioned False).

    obs_seq (B, h, 29) f64 with h <= obs_history given steps; prev_action (B, 8); has_prev (B,) bool (False = the
    reference's ``prev_actions is None``: raw zeros); goal_xy (B, 2) or (2,).  Returns float32 (B, 97)."""
    meta = ANT_META if meta is None else meta
    obs = np.array(obs_seq, dtype=np.float64, copy=True)
    B = obs.shape[0]
    position = obs[:, -1, :2].copy()                                                        # :74
    obs[..., 2:] = (obs[..., 2:] - meta["Observations_mean"]) / meta["Observations_std"]    # :77
    rot = q_to_rot6d(obs[..., 3:7])                                                         # :78-79 (normalised quaternion)
    obs = np.concatenate([obs[..., :3], rot, obs[..., 7:]], axis=-1)                        # :80
    cond = np.zeros((B, obs_history, obs.shape[-1]))
    pad = obs_history - obs.shape[1]
    if pad > 0:
        cond[:, pad:, :] = obs
    else:
        cond[:, :] = obs[:, -obs_history:, :]
    obs_cond = torch.from_numpy(cond)[..., 2:].flatten(start_dim=1).to(torch.float32)        # :107,:112
    act = np.zeros((B, 8))
    hp = np.asarray(has_prev, dtype=bool)
    act[hp] = (np.asarray(prev_action, dtype=np.float64)[hp] - meta["Actions_mean"]) / meta["Actions_std"]
    act_cond = torch.from_numpy(act).to(torch.float32)
    g = torch.tensor(np.broadcast_to(np.asarray(goal_xy, dtype=np.float64), (B, 2)) - position).float()
    yaw = torch.zeros(B, dtype=torch.float32)                                               # :82
    c, sn = torch.cos(yaw), torch.sin(yaw)
    rotm = torch.stack([torch.stack([c, sn], dim=1), torch.stack([-sn, c], dim=1)], dim=1)
    g = torch.tanh(torch.matmul(rotm, g.unsqueeze(2)).squeeze(2) / local_map_size)
    return torch.cat([obs_cond, act_cond, g], dim=1).numpy()


def get_timesteps(schedule: str, k_steps: int, exp_scale: float = 1.0):
    """common/fm_utils.py:4-17 (torch float32 arithmetic, as the reference)."""
    t = torch.linspace(0, 1, k_steps + 1)[:-1]
    if schedule == "linear":
        dt = torch.ones(k_steps) / k_steps
    elif schedule == "cosine":
        dt = torch.cos(t * torch.pi) + 1
        dt = dt / torch.sum(dt)
    elif schedule == "exp":
        dt = torch.exp(-t * exp_scale)
        dt = dt / torch.sum(dt)
    else:
        raise ValueError(f"Invalid schedule: {schedule}")
    t0 = torch.cat((torch.zeros(1), torch.cumsum(dt, dim=0)[:-1]))
    return t0, dt


def car_cond_vector(state, prev_action, has_prev, goal_xy, local_map_size=20, meta=CAR_META):
    """fm_policy.py:60-143 for carmaze, obs_history = action_history = 1.

    state (B, 6) f64 -- the last observation of the edge (== node state);
    prev_action (B, 2) f64 and has_prev (B,) bool -- ``prev_actions is None`` leaves
    raw zeros, *not* normalised (:113-122);  goal_xy (B, 2) or (2,) f64.
    Returns float32 (B, 7): [v, D, delta | a_prev(2) | goal(2)].
    """
    state = np.asarray(state, dtype=np.float64)
    B = state.shape[0]
    position = state[:, :2]
    yaw = state[:, 2]
    obs_n = (state - meta["Observations_mean"]) / meta["Observations_std"]       # :76
    obs_cond = torch.from_numpy(obs_n[:, 3:]).to(torch.float32)                   # :108,:110,:112
    act = np.zeros((B, 2))
    hp = np.asarray(has_prev, dtype=bool)
    act[hp] = (np.asarray(prev_action, dtype=np.float64)[hp] - meta["Actions_mean"]) / meta["Actions_std"]
    act_cond = torch.from_numpy(act).to(torch.float32)
    g = np.broadcast_to(np.asarray(goal_xy, dtype=np.float64), (B, 2)) - position  # :127
    g = torch.tensor(g).float()
    yaw_t = torch.tensor(yaw, dtype=torch.float32)
    c, s = torch.cos(yaw_t), torch.sin(yaw_t)
    rot = torch.stack([torch.stack([c, s], dim=1), torch.stack([-s, c], dim=1)], dim=1)
    g = torch.matmul(rot, g.unsqueeze(2)).squeeze(2)
    g = torch.tanh(g / local_map_size)                                            # :141-142
    return torch.cat([obs_cond, act_cond, g], dim=1).numpy()


def scale_local_map(local_map: np.ndarray) -> np.ndarray:
    """fm_policy.py:152."""
    return np.asarray(local_map, dtype=np.float32) * 2 - 1


def unnormalize_actions(naction_f32: np.ndarray, meta=CAR_META) -> np.ndarray:
    """fm_policy.py:201-203: float32 network output -> float64 actions."""
    return np.asarray(naction_f32, dtype=np.float32) * meta["Actions_std"] + meta["Actions_mean"]


@torch.no_grad()
def flow_sample(net, noise, local_map_scaled, cond, k_steps=1):
    """fm_policy.py:183-194 (flow_matching branch).  ``net(sample, local_map, t, cond)``."""
    x = torch.as_tensor(noise, dtype=torch.float32)
    lm = torch.as_tensor(local_map_scaled, dtype=torch.float32)
    cd = torch.as_tensor(cond, dtype=torch.float32)
    t0, dt = get_timesteps("exp", k_steps, exp_scale=4.0)
    for k in range(k_steps):
        ts = torch.ones((x.shape[0],)) * t0[k]
        ts = ts * 20
        v = net(sample=x, local_map=lm, timestep=ts, global_cond=cd)
        x = x.detach().clone() + v * dt[k]
    return x.numpy()
