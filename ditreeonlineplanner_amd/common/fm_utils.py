"""Flow-matching step schedule: start times ``t0`` and step sizes ``dt`` of the K Euler steps
(reference behaviour: common/fm_utils.py:4-17).  Host side, K values, evaluated in torch float32 so that
t0 / dt carry the same bits the reference sampler would use (checked against tests/golden/timesteps.json).
"""
import torch

_WEIGHTS = {
    # un-normalised step weights on the grid s_i = i / K, i < K
    "cosine": lambda s, scale: torch.cos(s * torch.pi) + 1,
    "exp": lambda s, scale: torch.exp(-s * scale),
}


def get_timesteps(schedule: str, k_steps: int, exp_scale: float = 1.0):
    grid = torch.linspace(0, 1, k_steps + 1)[:-1]
    if schedule == "linear":
        steps = torch.ones(k_steps) / k_steps
    elif schedule in _WEIGHTS:
        weights = _WEIGHTS[schedule](grid, exp_scale)
        steps = weights / torch.sum(weights)
    else:
        raise ValueError(f"Invalid schedule: {schedule}")
    starts = torch.cat((torch.zeros(1), torch.cumsum(steps, dim=0)[:-1]))
    return starts, steps
