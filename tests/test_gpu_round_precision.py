"""Round-level deviation of every denoiser instantiation (include/ditree.h DITREE_PREC_*) against the oracle round
(reference: planners/RRT.py:131-217 -- sample, nearest node, chunks of [local map, sampler, 8 env steps], accept).

Two workloads, both on the engine and on the CPU oracle (numpy f64 geometry + torch-CPU fp32 denoiser) with the same seeded
weights, samples and noise:

  "config2"  ONE round at BASELINE config 2's size: 1024 candidates against a 1024-node tree snapshot on boxes.csv, H = 32
             (4 chunks x [local map, conditioning, denoiser, 8 bicycle steps with goal / collision tests]), then accept.
  "growing"  THREE rounds of 256 candidates on a growing tree: the nodes accepted in round r are nearest-node candidates and
             parents of round r + 1, so a wrong accept shows up in every later parent index.

The tree snapshot keeps every node further than 3.5 cells from the goal (no candidate can end the plan inside the round:
round 2's test accepted 4 of 256 candidates because the lowest goal-reaching index ends the round), so more than half of the
candidates survive all four chunks and are appended: "tree parents bit-exact" rests on hundreds of nodes, not four.

Per instantiation and workload (gpurun_out/round_precision.json, committed as profiles/r03_round_precision.json):
  flips = candidates whose final status, number of chunks run, or executed steps of ANY chunk differ from the oracle's
  (a candidate that collides one step earlier is a flip, not a silent drop-out); n_agree = candidates - flips;
  max / median / 99th percentile of |d state| over the trajectories of the agreeing candidates; nearest-node mismatches;
  tree parent-index mismatches after every accept.
The north star asks for flags / parents exact and states within 1e-5: the f32 MFMA and the f16x3 instantiations are held to
that with n_agree == candidates; bf16x3 / f16 / bf16 are the throughput modes and are held to their measured deviation.

Map-sensitive candidates.  The local map (common/map_utils.py:391-459) is a DISCONTINUOUS function of the state: each of its
400 sample points is looked up in the occupancy grid by floor(), so a state that differs from the oracle's by 1e-6 reads a
different cell whenever a sample point lies within ~1e-6 of a cell boundary with different occupancy on the two sides -- and
the changed map changes the next denoiser call by ~1e-3.  With 400 points x 4 chunks x 256 candidates that happens to about
one candidate in a thousand for ANY arithmetic that is not bit-identical to the reference's (first seen on the f32 MFMA
instantiation: one collided candidate of 256 at 3.4e-4, all flags equal).  The oracle side of the test therefore computes, per
candidate, the smallest distance of any local-map sample point to an occupancy-changing cell boundary over its chunks
(`map_margin`); candidates with a margin below 1e-5 (the tolerance; about 1.5 % of the candidates) are reported as
`map_sensitive` and held to a sanity bound only, every other candidate to the tolerance."""
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
H, A, P, N0 = 32, 8, 64, 1024
WORKLOADS = {"config2": (1024, 1), "growing": (256, 3)}          # name -> (candidates per round, rounds)
# precision -> (max |d state| allowed, share of flipped candidates allowed, share of candidates that must stay within 1e-5,
# 99th percentile of |d state| allowed).  The pipeline is discontinuous (a state difference of 1e-5 can move a pose across a
# cell boundary of the local map, and the map conditions the next chunk's denoiser call), so below f32-class accuracy the
# MAXIMUM over the candidates is set by one or two outliers and moves by an order of magnitude with any change of summation
# order while median and 99th percentile stay put: the throughput modes are held to a multiple of their 99th percentile and
# their maximum only to a sanity bound.
BOUND = {1: (1e-5, 0.0, 1.0, 1e-5), 2: (1e-5, 0.0, 1.0, 1e-5), 3: (5e-3, 0.0, 0.7, 1e-4), 4: (1e-1, 0.01, 0.0, 1e-2),
         0: (2e-1, 0.02, 0.0, 7.5e-2)}
NAMES = {0: "bf16", 1: "f32", 2: "f16x3", 3: "bf16x3", 4: "f16"}


def map_margin(maze, states, n=20, scale=0.2, s_global=1.0):
    """(m, 6) chunk-start states -> (m,) distance [cells * s_global] from the nearest local-map sample point to a cell
    boundary across which the occupancy differs (inf when no such boundary is near any point).  Mirrors the sample-point
    construction of common/map_utils.py:391-459 (oracle.geometry.create_local_map)."""
    from oracle import geometry as G
    maze = np.asarray(maze)
    cx, cy = G.map_center(maze, 1.0)
    xs = G.local_axis(n, scale)
    xl, yl = np.meshgrid(xs, xs)
    xl, yl = xl.reshape(1, -1), yl.reshape(1, -1)
    c, s = np.cos(states[:, 2])[:, None], np.sin(states[:, 2])[:, None]
    u = (c * xl - s * yl + states[:, 0:1] + cx) / s_global
    w = (cy - (s * xl + c * yl + states[:, 1:2])) / s_global
    R, C = maze.shape
    ui, wi = np.floor(u).astype(np.int64), np.floor(w).astype(np.int64)
    here = maze[np.clip(wi, 0, R - 1), np.clip(ui, 0, C - 1)]
    out = np.full(states.shape[0], np.inf)
    for frac, idx, other in ((u - ui, ui, lambda d: maze[np.clip(wi, 0, R - 1), np.clip(ui + d, 0, C - 1)]),
                             (w - wi, wi, lambda d: maze[np.clip(wi + d, 0, R - 1), np.clip(ui, 0, C - 1)])):
        lo_diff = other(-1) != here               # crossing the lower boundary of the cell changes the value read
        hi_diff = other(+1) != here
        d = np.where(lo_diff, frac, np.inf)
        d = np.minimum(d, np.where(hi_diff, 1.0 - frac, np.inf))
        out = np.minimum(out, d.min(axis=1) * s_global)
    return out


def make_net():
    torch.manual_seed(0)
    onet = OD.init_noise_pred_net().eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in onet.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    return onet


def make_inputs(n_cand, seed):
    import bench
    maze = load_maze("boxes")
    nodes, goal, samples, cond, noise = bench.synth_inputs(maze, n_cand, seed=seed)
    # no node within 3.5 cells of the goal: 32 steps at v <= 4 move a car at most 2.6 cells, nobody reaches the goal radius
    near = np.hypot(nodes[:, 0] - goal[0], nodes[:, 1] - goal[1]) < 3.5
    far = np.nonzero(~near)[0]
    nodes[near] = nodes[far[np.arange(int(near.sum())) % len(far)]]
    return maze, nodes, goal, samples, cond, noise


def _oracle_planner(onet, maze, nodes, goal, noise):
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
    return pl


def oracle_refs(onet, name):
    """The oracle rounds of one workload: a pure function of the seeds above (cached under tests/golden/oracle_cache/)."""
    Bc, rounds = WORKLOADS[name]
    maze, nodes, goal, samples, cond, noise = make_inputs(Bc * rounds, 20260105 + rounds)
    pl = _oracle_planner(onet, maze, nodes, goal, noise)
    t = pl.tree
    refs = []
    for r in range(rounds):
        ref = pl.expand_round(samples[r * Bc:(r + 1) * Bc], cond[r * Bc:(r + 1) * Bc])
        assert pl.goal_node is None                       # the workload is built so that no round ends early
        ref["accepted"] = np.array(ref["accepted"], dtype=np.int64)
        ref["tree_parents"] = np.array(t.parents)
        ref["tree_states"] = np.array(t.states)
        run = np.arange(H // A)[None, :] < ref["chunks_run"][:, None]             # (B, n_chunks): chunks that ran
        mm = np.full(run.shape, np.inf)
        mm[run] = map_margin(maze, ref["states"][:, :, 0][run])
        ref["map_margin"] = mm.min(axis=1)
        refs.append(ref)
    return refs


@pytest.fixture(scope="module")
def cases():
    from tests.util import oracle_cache
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    onet = make_net()
    out = {}
    for name, (Bc, rounds) in WORKLOADS.items():
        maze, nodes, goal, samples, cond, noise = make_inputs(Bc * rounds, 20260105 + rounds)
        refs, cached = oracle_cache(f"round_precision_{name}", lambda: oracle_refs(onet, name))
        if cached:
            # live probe: the first candidates of round 0 re-computed now (same snapshot, same noise rows) must be the cached ones
            n = 6
            live = _oracle_planner(onet, maze, nodes, goal, noise).expand_round(samples[:n], cond[:n])
            assert np.array_equal(live["status"], refs[0]["status"][:n]) and np.array_equal(live["parent"], refs[0]["parent"][:n])
            # (torch-CPU convolutions differ by ~1e-6 between host CPUs: the probe catches a stale cache, not rounding)
            assert np.abs(live["states"] - refs[0]["states"][:n]).max() < 2e-5, "tests/golden/oracle_cache is stale: re-run make_oracle_cache.py"
        out[name] = dict(maze=maze, nodes=nodes, goal=goal, samples=samples, cond=cond, noise=noise, refs=refs, B=Bc, rounds=rounds)
    return onet, out


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def run_engine(ctx, onet, st, prec):
    from ditreeonlineplanner_amd.engine import CNT_GOAL, CNT_LATCH, CNT_NODES, ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    Bc, rounds = st["B"], st["rounds"]
    net = NoisePredNet(init=False)
    net.load_state_dict(onet.state_dict())
    net.bind(ctx, precision=prec, max_batch=Bc)
    eng = ExpansionEngine(ctx, st["maze"], st["nodes"][0], st["goal"], edge_length=H, action_horizon=A, pred_horizon=P,
                          batch=Bc, capacity=N0 + Bc * rounds, emulate_sticky_done=False)
    dev = ctx.device
    t = eng.tree
    nd = torch.as_tensor(st["nodes"], device=dev)
    t.state[:N0] = nd
    t.xy[:N0] = nd[:, :2]
    t.parent[:N0] = torch.arange(-1, N0 - 1, device=dev, dtype=torch.int32).clamp(min=0)
    t.parent[0] = -1
    t.has_prev[1:N0] = 1          # the root has no previous action (as the oracle tree)
    t.counters[CNT_NODES] = N0
    t.counters[CNT_GOAL] = -1
    t.counters[CNT_LATCH] = 0
    t.n_nodes_host = N0
    got = []
    for r in range(rounds):
        sl = slice(r * Bc, (r + 1) * Bc)
        eng.expand_round(torch.as_tensor(st["samples"][sl], device=dev), torch.as_tensor(st["cond"][sl], device=dev),
                         noise=st["noise"][sl].to(dev))
        rb = eng.rb
        snap = eng.tree_snapshot()
        got.append(dict(status=rb.status[:Bc].cpu().numpy() & 0xFF, parent=rb.parent[:Bc].cpu().numpy(),
                        end_state=rb.end_state[:Bc].cpu().numpy(), states=rb.states[:Bc].cpu().numpy(),
                        chunks_run=rb.chunks_run[:Bc].cpu().numpy(), chunk_steps=rb.chunk_steps[:Bc].cpu().numpy(),
                        tree_parents=snap["parents"], tree_states=snap["states"]))
    return got


def deviation(got, ref):
    """One round.  A candidate AGREES when its final status, its number of chunks and the executed steps of every chunk equal
    the oracle's; everything else is a flip (and its states are not compared: they belong to different trajectories)."""
    Bc = len(ref["status"])
    same_status = got["status"] == ref["status"]
    same_chunks = got["chunks_run"] == ref["chunks_run"]
    same_steps = (got["chunk_steps"] == ref["chunk_steps"]).all(axis=1)
    agree = same_status & same_chunks & same_steps
    sensitive = ref["map_margin"] < 1e-5
    per_cand, sens_dev = [], []
    for b in np.nonzero(agree)[0]:
        db = float(np.abs(got["end_state"][b] - ref["end_state"][b]).max())
        for j in range(int(ref["chunks_run"][b])):
            k = int(ref["chunk_steps"][b, j]) + 1
            db = max(db, float(np.abs(got["states"][b, j, :k] - ref["states"][b, j, :k]).max()))
        (sens_dev if sensitive[b] else per_cand).append(db)
    per_cand = np.array(per_cand) if per_cand else np.zeros(1)
    n = min(len(got["tree_parents"]), len(ref["tree_parents"]))
    tree_mis = int((got["tree_parents"][:n] != ref["tree_parents"][:n]).sum()) + abs(len(got["tree_parents"]) - len(ref["tree_parents"]))
    d_nodes = float(np.abs(got["tree_states"][:n] - ref["tree_states"][:n]).max())
    return dict(candidates=Bc, n_agree=int(agree.sum()), flips=int((~agree).sum()), status_flips=int((~same_status).sum()),
                chunks_run_mismatches=int((~same_chunks).sum()), chunk_steps_mismatches=int((~same_steps).sum()),
                nn_parent_mismatches=int((got["parent"] != ref["parent"]).sum()),
                map_sensitive=int(sensitive.sum()), max_abs_state_map_sensitive=float(max(sens_dev)) if sens_dev else 0.0,
                max_abs_trajectory_state=float(per_cand.max()), median_abs_trajectory_state=float(np.median(per_cand)),
                p99_abs_trajectory_state=float(np.quantile(per_cand, 0.99)), share_within_1e5=float((per_cand < 1e-5).mean()),
                tree_parent_mismatches=tree_mis, max_abs_tree_node_state=d_nodes,
                accepted_nodes_ref=int(len(ref["tree_parents"]) - N0), survived_all_chunks_ref=int((ref["status"] == 0).sum()),
                collided_ref=int((ref["status"] == 2).sum()))


def _record(prec, name, devs):
    out = os.path.join(REPO, "gpurun_out", "round_precision.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    allr = {}
    if os.path.exists(out):
        with open(out) as f:
            allr = json.load(f)
    allr.setdefault(NAMES[prec], {})[name] = devs
    with open(out, "w") as f:
        json.dump(allr, f, indent=1)


@pytest.mark.parametrize("name", ["config2", "growing"])
def test_workload_has_a_thick_accept(cases, name):
    """The evidence has to rest on many appended nodes: >= 30 % of every round's candidates are accepted by the oracle."""
    _, out = cases
    st = out[name]
    grown = N0
    for ref in st["refs"]:
        acc = len(ref["tree_parents"]) - grown
        grown = len(ref["tree_parents"])
        assert acc >= 0.3 * st["B"], (name, acc)
        assert (ref["status"] == 2).sum() >= 0.1 * st["B"]         # and collisions are exercised too
    if name == "growing":      # later rounds really hang nodes under nodes of earlier rounds
        assert (st["refs"][-1]["parent"] >= N0).sum() > 10


@pytest.mark.parametrize("name", ["config2", "growing"])
@pytest.mark.parametrize("prec", [1, 2, 3, 4, 0])
def test_round_deviation(ctx, cases, prec, name):
    onet, out = cases
    st = out[name]
    got = run_engine(ctx, onet, st, prec)
    devs = [deviation(g, r) for g, r in zip(got, st["refs"])]
    _record(prec, name, devs)
    print(NAMES[prec], name, devs)
    tol, flip_share, share, p99 = BOUND[prec]
    f32_class = prec in (1, 2)
    for r, dev in enumerate(devs):
        assert dev["nn_parent_mismatches"] == 0 or r > 0, dev          # round 0: the nearest node never depends on the denoiser
        if f32_class:
            # flags, chunk counts and the colliding / goal step of EVERY candidate, parents of every appended node: exact
            assert dev["n_agree"] == dev["candidates"] and dev["flips"] == 0, dev
            assert dev["nn_parent_mismatches"] == 0 and dev["tree_parent_mismatches"] == 0, dev
            assert dev["max_abs_tree_node_state"] < 1e-5, dev
        else:
            assert dev["flips"] <= flip_share * dev["candidates"], dev
            if r == 0 and dev["flips"] == 0:
                assert dev["tree_parent_mismatches"] == 0, dev
        assert dev["map_sensitive"] <= max(3, 0.03 * dev["candidates"]), dev     # the classification must stay the exception
        if r == 0 or f32_class:            # after a flip the trees differ: later rounds of a throughput mode are informational
            assert dev["max_abs_state_map_sensitive"] < max(tol, 5e-2), dev
            assert dev["max_abs_trajectory_state"] < tol, dev
            assert dev["share_within_1e5"] >= share, dev
            assert dev["p99_abs_trajectory_state"] < p99, dev
