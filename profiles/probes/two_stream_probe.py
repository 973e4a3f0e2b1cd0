"""Would two half-batches on two streams beat one batch on one stream?  (kernel gaps hidden, chip-level phases decorrelated)
Two contexts with their own weights / workspaces, B/2 candidates each, denoiser evaluations issued alternately on two
torch streams, against one context with B candidates.    python profiles/probes/two_stream_probe.py   (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ditreeonlineplanner_amd import _lib  # noqa: E402
from ditreeonlineplanner_amd.model import NoisePredNet  # noqa: E402
from ditreeonlineplanner_amd.ops import Context  # noqa: E402


def inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(B, 64, 2, generator=g).cuda()
    lm = (torch.rand(B, 20, 20, generator=g) > 0.7).float().mul(2).sub(1).cuda()
    cond = (torch.randn(B, 7, generator=g) * 0.7).cuda()
    return noise, lm, cond


def main():
    B, N = 1024, 24
    prec = _lib.PREC_NAMES[sys.argv[1]] if len(sys.argv) > 1 else _lib.PREC_F16X3
    torch.manual_seed(0)
    one = Context(0)
    net = NoisePredNet()
    net.bind(one, precision=prec, max_batch=B)
    x = inputs(B, 1)
    for _ in range(3):
        one.denoise_eval(*x, 0.3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(N):
        one.denoise_eval(*x, 0.3)
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t) / N
    print(f"one stream, B = {B}: {t_one * 1e3:.3f} ms per evaluation")
    for parts in (2, 4):
        ctxs, ins, streams = [], [], []
        for i in range(parts):
            c = Context(0)
            n = NoisePredNet()
            n.load_state_dict(net.state_dict())
            n.bind(c, precision=prec, max_batch=B // parts)
            ctxs.append(c)
            ins.append(inputs(B // parts, 2 + i))
            streams.append(torch.cuda.Stream())
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                for _ in range(3):
                    ctxs[i].denoise_eval(*ins[i], 0.3)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(N):
            for i in range(parts):
                with torch.cuda.stream(streams[i]):
                    ctxs[i].denoise_eval(*ins[i], 0.3)
        torch.cuda.synchronize()
        t_p = (time.perf_counter() - t) / N
        print(f"{parts} streams, B = {B // parts} each: {t_p * 1e3:.3f} ms per {B} candidates ({t_one / t_p:.3f}x)")
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
