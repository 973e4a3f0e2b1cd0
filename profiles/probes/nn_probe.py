"""Duration of ditree_nn_argmin (1024 queries) against the tree size: hipEvents, median of 20.  Writes gpurun_out/<name>.json."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from ditreeonlineplanner_amd.ops import Context            # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
res = {}
q = torch.as_tensor(rng.uniform(-10, 10, (1024, 2)), device="cuda")
for N in (1024, 16384, 131072, 1048576):
    nodes = torch.as_tensor(rng.uniform(-10, 10, (N, 2)), device="cuda")
    for _ in range(3):
        idx = ctx.nn_argmin(q, nodes)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record()
        idx = ctx.nn_argmin(q, nodes)
        b.record()
    torch.cuda.synchronize()
    ref = torch.cdist(q, nodes).argmin(dim=1)
    assert bool((idx.long() == ref).all())
    res[N] = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[10]
    print(N, res[N], flush=True)
name = sys.argv[1] if len(sys.argv) > 1 else "nn_probe"
with open(os.path.join(REPO, "gpurun_out", name + ".json"), "w") as f:
    json.dump({"what": "nn_argmin_kernel, 1024 queries, median us of 20 by number of nodes", "us_by_nodes": res}, f, indent=1)
