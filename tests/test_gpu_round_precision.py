"""Round-level deviation of every denoiser instantiation (include/ditree.h DITREE_PREC_*) against the oracle round.

One BASELINE-config-2-shaped round -- 256 candidates against a 1024-node tree snapshot on boxes.csv, H = 32
(4 chunks x [local map, conditioning, denoiser, 8 bicycle steps with goal / collision tests]) -- on the engine and
on the CPU oracle (numpy f64 geometry + torch-CPU fp32 denoiser), same seeded weights, samples and noise.  Recorded per
instantiation (gpurun_out/round_precision.json, committed as profiles/r02_round_precision.json):
  max |d state| over the trajectories of candidates whose status agrees, status flips, nearest-node parent mismatches,
  tree parent-index mismatches after accept.
The north star asks for flags / parents exact and states within 1e-5: the f32 MFMA and the f16x3 instantiations are held
to that; bf16x3 / f16 / bf16 are the throughput modes and are held to their measured deviation (x2)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import rrt as ORRT
from oracle import sampler as OS
from tests.util import REPO, load_maze

pytestmark = pytest.mark.gpu
B, H, A, P, N0 = 256, 32, 8, 64, 1024
# precision -> (max |d state| allowed, status flips allowed among the 256 candidates, share of candidates that must stay
# within 1e-5, 99th percentile of |d state| allowed).  Measured (profiles/r02_round_precision.json): f32 2.2e-6, f16x3 2.6e-6
# at the maximum, no flips.  The pipeline is discontinuous (a state difference of 1e-5 can move a pose across a cell boundary
# of the local map, and the map conditions the next chunk's denoiser call), so below f32-class accuracy the MAXIMUM over the
# candidates is set by one or two outliers and moves by an order of magnitude with any change of summation order (bf16x3:
# 1.1e-3 on one build, 2.4e-5 on the next; f16: 3.2e-3, then 2.8e-2) while median and 99th percentile stay put (bf16x3
# 2.7e-6 / 2.3e-5, f16 2.3e-4 / 3e-3, bf16 1.2e-3 / 2.4e-2).  The throughput modes are therefore held to three times their
# 99th percentile, and their maximum only to a sanity bound.
BOUND = {1: (1e-5, 0, 1.0, 1e-5), 2: (1e-5, 0, 1.0, 1e-5), 3: (5e-3, 0, 0.7, 1e-4), 4: (1e-1, 2, 0.0, 1e-2),
         0: (2e-1, 4, 0.0, 7.5e-2)}
NAMES = {0: "bf16", 1: "f32", 2: "f16x3", 3: "bf16x3", 4: "f16"}


@pytest.fixture(scope="module")
def setup():
    import bench
    maze = load_maze("boxes")
    nodes, goal, samples, cond, noise = bench.synth_inputs(maze, B, seed=20260105)
    torch.manual_seed(0)
    onet = OD.init_noise_pred_net().eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in onet.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    nz = noise.numpy()

    def sampler(cand_idx, chunk, state, prev_action, has_prev, cond_goal, local_map):
        cv = OS.car_cond_vector(state, prev_action, has_prev, cond_goal)
        x1 = OS.flow_sample(onet, nz[cand_idx, chunk], OS.scale_local_map(local_map), cv, k_steps=1)
        return OS.unnormalize_actions(x1)

    pl = ORRT.OraclePlanner(maze, nodes[0], goal, sampler, edge_length=H, action_horizon=A, emulate_sticky_done=False)
    t = pl.tree
    for i in range(1, len(nodes)):
        t.states.append(nodes[i].copy()); t.parents.append(max(i - 1, 0)); t.last_action.append(np.zeros(2))
        t.has_prev.append(True); t.num_visit.append(0); t.edge_states.append(None); t.edge_actions.append(None)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = pl.expand_round(samples, cond)
    ref["tree_parents"] = np.array(t.parents)
    return dict(maze=maze, nodes=nodes, goal=goal, samples=samples, cond=cond, noise=noise, onet=onet, ref=ref)


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def run_engine(ctx, st, prec):
    from ditreeonlineplanner_amd.engine import CNT_GOAL, CNT_LATCH, CNT_NODES, ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    net = NoisePredNet()
    net.load_state_dict(st["onet"].state_dict())
    net.bind(ctx, precision=prec, max_batch=B)
    eng = ExpansionEngine(ctx, st["maze"], st["nodes"][0], st["goal"], edge_length=H, action_horizon=A, pred_horizon=P,
                          batch=B, capacity=N0 + B, emulate_sticky_done=False)
    dev = ctx.device
    t = eng.tree
    nd = torch.as_tensor(st["nodes"], device=dev)
    t.state[:N0] = nd
    t.xy[:N0] = nd[:, :2]
    t.parent[:N0] = torch.arange(-1, N0 - 1, device=dev, dtype=torch.int32).clamp(min=0)
    t.parent[0] = -1
    t.has_prev[:N0] = 1
    t.counters[CNT_NODES] = N0
    t.counters[CNT_GOAL] = -1
    t.counters[CNT_LATCH] = 0
    t.n_nodes_host = N0
    eng.expand_round(torch.as_tensor(st["samples"], device=dev), torch.as_tensor(st["cond"], device=dev),
                     noise=st["noise"].to(dev))
    rb = eng.rb
    return dict(status=rb.status[:B].cpu().numpy() & 0xFF, parent=rb.parent[:B].cpu().numpy(),
                end_state=rb.end_state[:B].cpu().numpy(), states=rb.states[:B].cpu().numpy(),
                chunks_run=rb.chunks_run[:B].cpu().numpy(), tree_parents=eng.tree_snapshot()["parents"])


def deviation(got, ref):
    same = got["status"] == ref["status"]
    flips = int((~same).sum())
    agree = same & (got["chunks_run"] == ref["chunks_run"])
    d_end = float(np.abs(got["end_state"][agree] - ref["end_state"][agree]).max()) if agree.any() else 0.0
    d_traj = 0.0
    per_cand = []
    for b in np.nonzero(agree)[0]:
        n = int(ref["chunks_run"][b])
        live = ref["chunk_steps"][b, :n]
        db = 0.0
        for j in range(n):
            k = int(live[j]) + 1
            db = max(db, float(np.abs(got["states"][b, j, :k] - ref["states"][b, j, :k]).max()))
        per_cand.append(db)
        d_traj = max(d_traj, db)
    per_cand = np.array(per_cand) if per_cand else np.zeros(1)
    n = min(len(got["tree_parents"]), len(ref["tree_parents"]))
    tree_mis = int((got["tree_parents"][:n] != ref["tree_parents"][:n]).sum()) + abs(len(got["tree_parents"]) - len(ref["tree_parents"]))
    return dict(status_flips=flips, nn_parent_mismatches=int((got["parent"] != ref["parent"]).sum()),
                max_abs_end_state=d_end, max_abs_trajectory_state=d_traj, tree_parent_mismatches=tree_mis,
                median_abs_trajectory_state=float(np.median(per_cand)), p99_abs_trajectory_state=float(np.quantile(per_cand, 0.99)),
                share_within_1e5=float((per_cand < 1e-5).mean()),
                candidates=B, accepted_nodes_ref=int(len(ref["tree_parents"]) - N0))


@pytest.mark.parametrize("prec", [1, 2, 3, 4, 0])
def test_round_deviation(ctx, setup, prec):
    got = run_engine(ctx, setup, prec)
    dev = deviation(got, setup["ref"])
    out = os.path.join(REPO, "gpurun_out", "round_precision.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    allr = {}
    if os.path.exists(out):
        with open(out) as f:
            allr = json.load(f)
    allr[NAMES[prec]] = dev
    with open(out, "w") as f:
        json.dump(allr, f, indent=1)
    print(NAMES[prec], dev)
    tol, flips, share, p99 = BOUND[prec]
    assert dev["nn_parent_mismatches"] == 0                      # nearest node never depends on the denoiser
    assert dev["status_flips"] <= flips, dev
    assert max(dev["max_abs_end_state"], dev["max_abs_trajectory_state"]) < tol, dev
    assert dev["share_within_1e5"] >= share, dev
    assert dev["p99_abs_trajectory_state"] < p99, dev
    if flips == 0:
        assert dev["tree_parent_mismatches"] == 0, dev
