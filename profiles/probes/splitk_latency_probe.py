"""Latency mode of the denoiser (DITREE_DENOISE_SPLITK=1: split-K of the halo kernel for small batches): duration of one denoiser
call (hipEvents, median of 20) with and without it at B = 1 ... 1024, and the largest difference of the outputs (the split sum is
a different, deterministic summation order).  f16x3, the car network, seeded random weights.  Writes gpurun_out/<name>.json."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from ditreeonlineplanner_amd.model import NoisePredNet     # noqa: E402
from ditreeonlineplanner_amd.ops import Context            # noqa: E402

torch.manual_seed(0)
ctx = Context(0)
net = NoisePredNet().eval()
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for n, p in net.named_parameters():
        if p.dim() == 1:
            p.add_(0.2 * torch.randn(p.shape, generator=g))
net.bind(ctx, precision=2, max_batch=1024)
unit = np.concatenate([np.zeros(2), np.ones(2)])
res = {}
for B in (1, 16, 64, 128, 256, 512, 1024):
    gi = torch.Generator().manual_seed(100 + B)
    noise = torch.randn(B, 64, 2, generator=gi).cuda()
    lm = ((torch.rand(B, 20, 20, generator=gi) < 0.3).float() * 2 - 1).cuda()
    cond = (torch.randn(B, 7, generator=gi) * 0.6).cuda()
    out, t = {}, {}
    for mode in ("0", "1"):
        os.environ["DITREE_DENOISE_SPLITK"] = mode
        for _ in range(3):
            x = ctx.denoise(noise, lm, cond, act_norm=unit, want_actions=False)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record()
            x = ctx.denoise(noise, lm, cond, act_norm=unit, want_actions=False)
            b.record()
        torch.cuda.synchronize()
        t[mode] = sorted(a.elapsed_time(b) for a, b in ev)[10]
        out[mode] = x.cpu().numpy().astype(np.float64)
        x2 = ctx.denoise(noise, lm, cond, act_norm=unit, want_actions=False).cpu().numpy()
        assert np.array_equal(x2, out[mode].astype(np.float32)), "not deterministic"
    d = np.abs(out["1"] - out["0"]).max() / max(np.sqrt(np.mean(out["0"] ** 2)), 1e-30)
    res[B] = {"ms_plain": t["0"], "ms_splitk": t["1"], "speedup": t["0"] / t["1"], "max_abs_diff_over_rms": d}
    print(B, res[B], flush=True)
os.environ["DITREE_DENOISE_SPLITK"] = "0"
name = sys.argv[1] if len(sys.argv) > 1 else "splitk_latency_probe"
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
with open(os.path.join(REPO, "gpurun_out", name + ".json"), "w") as f:
    json.dump({"what": "one denoiser call, f16x3, car network: median ms of 20 calls, plain vs DITREE_DENOISE_SPLITK=1", "by_batch": res}, f, indent=1)
