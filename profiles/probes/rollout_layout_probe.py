"""Durations of the rollout kernels by output layout (hipEvents on the launch stream, 30 launches each after 5 warm-ups):
car_rollout_kernel 65 536 x 16 with (a) packed (17, 6) state rows + (16, 2) action rows per candidate, (b) the same rows stored
step-major / component-major / candidate-minor, (c) action rows only, (d) no rows; ant_rollout_kernel<model> 65 536 x 16 with
packed rows, candidate-minor rows, no rows.  Writes gpurun_out/rollout_layout_probe.json."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from ditreeonlineplanner_amd import _lib                      # noqa: E402
from ditreeonlineplanner_amd._lib import check, lib          # noqa: E402
from ditreeonlineplanner_amd.ops import Context, _dbl, _ptr   # noqa: E402

K, T = 65536, 16
maze = np.loadtxt(os.path.join(REPO, "ditreeonlineplanner_amd", "data", "boxes.csv"), delimiter=",")
ctx = Context(0)
ctx.upload_maze(maze)
rng = np.random.default_rng(1)
free = np.argwhere(maze[1:-1, 1:-1] == 0) + 1
cell = free[rng.integers(0, len(free), K)]
st0 = np.stack([(cell[:, 1] + 0.5) - 10 + rng.uniform(-0.25, 0.25, K), 10 - (cell[:, 0] + 0.5) + rng.uniform(-0.25, 0.25, K),
                rng.uniform(-np.pi, np.pi, K), rng.uniform(0, 4, K), rng.uniform(0, 1, K), rng.uniform(-0.4, 0.4, K)], axis=1)
s0 = torch.as_tensor(st0, device="cuda")
act = torch.as_tensor(np.stack([rng.normal(0.45, 1.0, (K, T)), rng.normal(0.0, 0.92, (K, T))], axis=2).copy(), device="cuda")
g, gp = _dbl(np.array([7.5, 7.5]))


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        fn(a, b)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return {"median_us": t[len(t) // 2], "min_us": t[0], "max_us": t[-1]}


res = {"K": K, "T": T, "car": {}, "ant_model": {}}
steps = torch.zeros(K, dtype=torch.int32, device="cuda")
f64 = torch.float64
variants = {
    "rows_packed": (torch.zeros(K, T + 1, 6, dtype=f64, device="cuda"), torch.zeros(K, T, 2, dtype=f64, device="cuda")),
    "rows_candidate_minor": (torch.zeros(T + 1, 6, K, dtype=f64, device="cuda").permute(2, 0, 1), torch.zeros(T, 2, K, dtype=f64, device="cuda").permute(2, 0, 1)),
    "action_rows_only_packed": (None, torch.zeros(K, T, 2, dtype=f64, device="cuda")),
    "no_rows": (None, None),
}
for name, (so, ao) in variants.items():
    state = s0.clone()
    status = torch.zeros(K, dtype=torch.int32, device="cuda")
    sl = _lib.Strides(*so.stride()) if so is not None else None
    al = _lib.Strides(*ao.stride()) if ao is not None else None

    def fn(e0=None, e1=None):
        state.copy_(s0)
        status.zero_()
        if e0 is not None:
            e0.record()
        check(ctx._h, lib().ditree_car_rollout_ld(ctx._h, _ptr(state), _ptr(act), T * 2, _ptr(status), K, T, gp, _ptr(so),
                                                  C.byref(sl) if sl else None, _ptr(ao), C.byref(al) if al else None, _ptr(steps),
                                                  None, None, ctx.stream), "car_rollout")
        if e1 is not None:
            e1.record()
    res["car"][name] = timed(fn)
    res["car"][name]["algorithmic_MB"] = (K * (48 + 16 * T) + (K * 48 * (T + 1) if so is not None else 0) + (K * 16 * T if ao is not None else 0) + K * 56) / 1e6

# ant stand-in model (NOT MuJoCo)
sa = np.zeros((K, 29))
sa[:, 0] = st0[:, 0] * 4.0
sa[:, 1] = st0[:, 1] * 4.0
sa[:, 2], sa[:, 3] = 0.75, 1.0
sa[:, 7:15] = np.tile([0.0, 0.87], 4)
sa0 = torch.as_tensor(sa, device="cuda")
aact = torch.as_tensor(np.clip(rng.uniform(-1, 1, (K, 1, 8)) + rng.normal(0, 0.4, (K, T, 8)), -1.2, 1.2), device="cuda")
model = _lib.AntModel.default()
gd, gdp = _dbl(np.array([30.0, 30.0]))
avariants = {
    "rows_packed": (torch.zeros(K, T + 1, 29, dtype=f64, device="cuda"), torch.zeros(K, T, 8, dtype=f64, device="cuda")),
    "rows_candidate_minor": (torch.zeros(T + 1, 29, K, dtype=f64, device="cuda").permute(2, 0, 1), torch.zeros(T, 8, K, dtype=f64, device="cuda").permute(2, 0, 1)),
    "no_rows": (None, None),
}
for name, (so, ao) in avariants.items():
    state = sa0.clone()
    status = torch.zeros(K, dtype=torch.int32, device="cuda")
    sl = _lib.Strides(*so.stride()) if so is not None else None
    al = _lib.Strides(*ao.stride()) if ao is not None else None

    def fn(e0=None, e1=None):
        state.copy_(sa0)
        status.zero_()
        if e0 is not None:
            e0.record()
        check(ctx._h, lib().ditree_ant_rollout(ctx._h, C.byref(model), _ptr(state), _ptr(aact), T * 8, None, 0, _ptr(status), K, T, gdp,
                                               1.8, 1.2, 4.0, _ptr(so), C.byref(sl) if sl else None, _ptr(ao),
                                               C.byref(al) if al else None, _ptr(steps), ctx.stream), "ant_rollout")
        if e1 is not None:
            e1.record()
    res["ant_model"][name] = timed(fn)
    res["ant_model"][name]["algorithmic_MB"] = (K * (232 + 64 * T) + (K * 232 * (T + 1) if so is not None else 0) + (K * 64 * T if ao is not None else 0) + K * 240) / 1e6
    res["ant_model"][name]["collided"] = int((status == 2).sum().item())
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
res["car_kernel_variant"] = "software-pipelined steps (DITREE_ROLLOUT_PIPELINE=1)" if os.environ.get("DITREE_ROLLOUT_PIPELINE") == "1" else "default"
name = sys.argv[1] if len(sys.argv) > 1 else "rollout_layout_probe"
with open(os.path.join(REPO, "gpurun_out", name + ".json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
