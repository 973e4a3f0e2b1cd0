"""Where does NoisePredNet.bind spend its time?  (the GPU suite binds ~60 times)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ditreeonlineplanner_amd.model import NoisePredNet
from ditreeonlineplanner_amd.ops import Context
from ditreeonlineplanner_amd.weights import pack_state_dict
ctx = Context(0)
t = time.perf_counter(); net = NoisePredNet(seed=0); print("construct", round(time.perf_counter() - t, 3))
for rep in range(2):
    t = time.perf_counter(); blob, man = pack_state_dict(net.state_dict(), pred_horizon=64, local_map_size=20, checksum=False); print("pack", round(time.perf_counter() - t, 3))
    t = time.perf_counter(); ctx.load_weights(blob, man); print("load_weights", round(time.perf_counter() - t, 3))
    for prec in (2, 0, 1, 2):
        t = time.perf_counter(); ctx.denoise_reserve(64, prec); torch.cuda.synchronize(); print("reserve prec", prec, round(time.perf_counter() - t, 3))
    t = time.perf_counter(); ctx.denoise_reserve(1024, 2); torch.cuda.synchronize(); print("reserve 1024", round(time.perf_counter() - t, 3))
