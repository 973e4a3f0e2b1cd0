"""Flow-matching step schedule (mirror of the reference's common/fm_utils.py:4-17).

Host-side and tiny (K values); evaluated with torch float32 exactly like the reference
so that t0/dt are bit-identical to what the reference sampler would use.
"""
import torch


def get_timesteps(schedule: str, k_steps: int, exp_scale: float = 1.0):
    t = torch.linspace(0, 1, k_steps + 1)[:-1]
    if schedule == "linear":
        dt = torch.full((k_steps,), 1.0) / k_steps
    elif schedule == "cosine":
        dt = torch.cos(t * torch.pi) + 1
        dt = dt / dt.sum()
    elif schedule == "exp":
        dt = torch.exp(-t * exp_scale)
        dt = dt / dt.sum()
    else:
        raise ValueError(f"Invalid schedule: {schedule}")
    t0 = torch.cat((torch.zeros(1), torch.cumsum(dt, dim=0)[:-1]))
    return t0, dt
