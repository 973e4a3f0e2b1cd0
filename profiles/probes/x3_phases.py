"""Issue time of the six MFMA phases of a K-step in the split halo kernel (diagnostic build: denoise_kernels.hip compiled
with -DHALO16_STAMP -DX3_PHASE_STAMP).  Every wave of work-group 100 of a Cin = 2048 launch stamps s_memtime (core clock)
at the head of each phase of its last three in-loop K-steps (taps 0, 1, 2 of one channel chunk):
    0 P1  1 P2  2 P3  3 P4  4 P5  5 wait+barrier  6 reads + P6 (LDS-DMA issue)  7 end
An ideal phase is 16 MFMAs x 16 cycles x 2 waves of the SIMD = 512 cycles.
    python profiles/probes/x3_phases.py            (on the GPU box)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ditreeonlineplanner_amd import _lib  # noqa: E402
from ditreeonlineplanner_amd.model import NoisePredNet  # noqa: E402
from ditreeonlineplanner_amd.ops import Context  # noqa: E402


def main():
    B = 1024
    h = _lib.lib()
    ctx = Context(0)
    torch.manual_seed(0)
    net = NoisePredNet()
    net.bind(ctx, precision=_lib.PREC_F16X3, max_batch=B)
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(B, 64, 2, generator=g).cuda()
    lm = (torch.rand(B, 20, 20, generator=g) > 0.7).float().mul(2).sub(1).cuda()
    cond = (torch.randn(B, 7, generator=g) * 0.7).cuda()
    for _ in range(3):
        ctx.denoise_eval(noise, lm, cond, 0.3)
    torch.cuda.synchronize()
    buf = np.zeros((8, 32), dtype=np.uint32)
    fn = h.ditree_debug_x3_phase
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p]
    assert fn(buf.ctypes.data) == 0
    names = ["P1", "P2", "P3", "P4", "P5", "wait+bar", "rd+P6"]
    print("nv", buf[:, 24])
    st = buf[:, :24].astype(np.int64).reshape(8, 3, 8)
    t0 = st[:, 0, 0].min()
    print("wave  tap   " + "  ".join(f"{n:>8s}" for n in names) + "     step   start-offset")
    for w in range(8):
        for T in range(3):
            d = np.diff(st[w, T])
            print(f"{w:4d} {T:4d}   " + "  ".join(f"{int(x):8d}" for x in d) + f"  {int(st[w, T, 7] - st[w, T, 0]):7d}   {int(st[w, T, 0] - t0):7d}")
    d = np.diff(st, axis=2).reshape(24, 7)
    print("mean       " + "  ".join(f"{x:8.0f}" for x in d.mean(0)) + f"  {d.sum(1).mean():7.0f}")
    # step-to-step period: tap 1 start - tap 0 start
    print("period tap0->tap1, tap1->tap2 per wave:", (st[:, 1, 0] - st[:, 0, 0]).tolist(), (st[:, 2, 0] - st[:, 1, 0]).tolist())


if __name__ == "__main__":
    main()
