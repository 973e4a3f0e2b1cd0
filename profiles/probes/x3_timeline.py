"""In-kernel timeline of the split halo kernel (diagnostic build: denoise_kernels.hip compiled with -DHALO16_STAMP).

    hipcc ... -DHALO16_STAMP -c denoise_kernels.hip && python -m ditreeonlineplanner_amd.build
    python profiles/probes/x3_timeline.py [B] [out.json]            (on the GPU box)

Every work-group of conv3_halo16x3_kernel records {start, loop start, epilogue start, end} on the 100 MHz real-time
clock plus HW_ID / XCC_ID.  Per launch (one denoiser evaluation at B candidates, f16x3) the script prints the
median prologue / K loop / epilogue, the launch span, and how the 256 tiles were spread over the CUs."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ditreeonlineplanner_amd import _lib  # noqa: E402
from ditreeonlineplanner_amd.model import NoisePredNet  # noqa: E402
from ditreeonlineplanner_amd.ops import Context  # noqa: E402

MAXREC = 65536


def read(h, reset):
    buf = np.zeros((MAXREC, 6), dtype=np.uint64)
    cnt = C.c_uint32(0)
    fn = h.ditree_debug_x3_stamp
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    rc = fn(buf.ctypes.data, C.byref(cnt), int(reset))
    assert rc == 0, rc
    return buf[:min(cnt.value, MAXREC)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/x3_timeline.json"
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
    read(h, True)
    ctx.denoise_eval(noise, lm, cond, 0.3)
    torch.cuda.synchronize()
    rec = read(h, True)
    t = rec[:, :4].astype(np.int64)
    nv = (rec[:, 4] >> np.uint64(32)).astype(np.int64)
    hw = (rec[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (rec[:, 5] >> np.uint64(32)).astype(np.int64) & 0xf
    # launches: records are appended in completion order; a launch = a run of records whose start times are apart
    # from the previous launch's end.  Sort by start and cut where start > running max end.
    order = np.argsort(t[:, 0], kind="stable")
    t, nv, hw, xcc = t[order], nv[order], hw[order], xcc[order]
    launches = []
    lo = 0
    end = t[0, 3]
    for i in range(1, len(t) + 1):
        if i == len(t) or t[i, 0] >= end:
            launches.append((lo, i))
            lo = i
        if i < len(t):
            end = max(end, t[i, 3])
    rows = []
    print(f"{len(t)} work-groups, {len(launches)} launches (times in us)")
    print("  nv  wgs  span   pro   loop   epi   us/step  CUs  2-on-a-CU  start-spread")
    for lo, hi in launches:
        tt = t[lo:hi]
        # gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        cu = (xcc[lo:hi] << 16) | (hw[lo:hi] & 0xff00)
        ids, counts = np.unique(cu, return_counts=True)
        span = (tt[:, 3].max() - tt[:, 0].min()) / 100.0
        pro = np.median(tt[:, 1] - tt[:, 0]) / 100.0
        loop = np.median(tt[:, 2] - tt[:, 1]) / 100.0
        epi = np.median(tt[:, 3] - tt[:, 2]) / 100.0
        steps = 3 * int(nv[lo])
        row = dict(nv=int(nv[lo]), wgs=int(hi - lo), span_us=span, prologue_us=pro, loop_us=loop, epilogue_us=epi,
                   loop_us_per_step=loop / steps, cus=int(len(ids)), cus_with_2=int((counts >= 2).sum()),
                   start_spread_us=(tt[:, 0].max() - tt[:, 0].min()) / 100.0)
        rows.append(row)
        print(f"{row['nv']:4d} {row['wgs']:4d} {span:6.1f} {pro:5.1f} {loop:6.1f} {epi:5.1f}   {loop / steps:6.3f}  "
              f"{row['cus']:4d}  {row['cus_with_2']:4d}     {row['start_spread_us']:6.1f}")
    tot = sum(r["span_us"] for r in rows)
    print(f"sum of spans {tot:.0f} us: prologue {sum(r['prologue_us'] for r in rows):.0f}, "
          f"loop {sum(r['loop_us'] for r in rows):.0f}, epilogue {sum(r['epilogue_us'] for r in rows):.0f}")
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(dict(B=B, launches=rows), f, indent=1)


if __name__ == "__main__":
    main()
