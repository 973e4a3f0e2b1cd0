"""Per-layer error of every denoiser instantiation against the torch-CPU fp32 oracle (B = 24, every layer tapped):
relative L2, max |err| / rms(ref), and the largest element-wise excess over atol*rms + rtol*|ref|.
    python profiles/denoiser_precision_report.py [out.json]       (on the GPU box)
The bounds of tests/test_gpu_denoiser.py are set at about twice these numbers."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditreeonlineplanner_amd import _lib  # noqa: E402
from ditreeonlineplanner_amd.model import NoisePredNet  # noqa: E402
from ditreeonlineplanner_amd.ops import Context  # noqa: E402
from tests import test_gpu_denoiser as T  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/denoiser_precision_report.json"
    onet = T._make_oracle_net()
    noise, lm, cond = T._make_inputs()
    B = noise.shape[0]
    x_ref, taps = T._oracle_with_taps(onet, noise, lm, cond)
    ctx = Context(0)
    report = {}
    for name, prec in _lib.PREC_NAMES.items():
        net = NoisePredNet()
        net.load_state_dict(onet.state_dict())
        net.bind(ctx, precision=prec, max_batch=B)
        x1 = ctx.denoise(noise.cuda(), lm.cuda(), cond.cuda(), want_actions=False).cpu().numpy()
        rep = {}
        for lname, _ in T.LAYERS:
            if lname == "enc.pool":
                continue
            got = ctx.debug_read(lname, B).cpu().numpy()
            rep[lname] = T.err_stats(got, T.tap_to_blc(taps[lname].numpy(), B))
        rep["x1"] = T.err_stats(x1, x_ref)
        report[name] = rep
        print(name, "x1", rep["x1"], "worst layer rel", max(v["rel_l2"] for v in rep.values()), flush=True)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
