"""Where do the 227 MB of car_rollout_kernel's 71 MB workload come from?  Three launches of the same 65 536 x 16 rollouts:
(a) all outputs (states (17, 6) + actions (16, 2) rows per candidate), (b) no state rows, (c) no state and no action rows.
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` (separate passes); profiles/probes/rollout_traffic_read.py
prints the per-launch counters in launch order."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
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
states = torch.zeros(K, T + 1, 6, dtype=torch.float64, device="cuda")
aout = torch.zeros(K, T, 2, dtype=torch.float64, device="cuda")
steps = torch.zeros(K, dtype=torch.int32, device="cuda")
g, gp = _dbl(np.array([7.5, 7.5]))
for so, ao in ((states, aout), (None, aout), (None, None)):
    state = s0.clone()
    status = torch.zeros(K, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    check(ctx._h, lib().ditree_car_rollout(ctx._h, _ptr(state), _ptr(act), T * 2, _ptr(status), K, T, gp, _ptr(so), (T + 1) * 6,
                                           _ptr(ao), T * 2, _ptr(steps), None, None, ctx.stream), "car_rollout")
    torch.cuda.synchronize()
print("done")
