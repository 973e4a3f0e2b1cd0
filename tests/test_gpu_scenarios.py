"""Scenario-level parity WITH THE DENOISER IN THE LOOP (the north star's wording: "outputs match the CPU reference per
scenario -- reached-goal flag and tree parent indices bit-exact, propagated states within 1e-5").

The golden traces pin the planner mechanics with tape actions; tests/test_gpu_round_precision.py pins single rounds with the
network.  Here whole PLANS run on both sides: scenarios of experiments/test_scenarios_car.csv (the rows the golden traces
use), `prop_duration = [64]` (8 chunks of 8 steps, the reference's car setting), rounds of 64 candidates drawn in the
reference's RNG order (seed 42), a candidate budget instead of the wall clock, the same seeded weights and noise -- engine
(f16x3, the default instantiation) vs oracle planner (numpy f64 geometry + torch-CPU fp32 denoiser).

Two comparisons per scenario:
  FREE-RUNNING  both planners grow their own trees.  The discrete outputs must be EQUAL: every parent index (a single flipped
                flag anywhere would change every later one), the reached flag, chunk iterations, candidates, the chosen node.
                Node STATES cannot stay within 1e-5 along a whole plan for any arithmetic that is not bit-identical to the
                reference's: a child starts from its parent's state, so the per-edge deviation (~1e-6) is carried and amplified
                down every chain of edges, and each of the ~3000 denoiser calls of a plan reads a local map that is a
                discontinuous function of that state (about 1 call in 400 has a sample point within 1e-5 of an occupancy
                boundary: tests/test_gpu_round_precision.py "map-sensitive").  The deviation is recorded
                (gpurun_out/scenario_parity.json) and bounded loosely.
  PER ROUND     the oracle expands every round from the ENGINE's tree (same node states, same previous actions): the
                deviation of one expansion without what the chain carried in.  Asserted: no flipped status / chunk count /
                colliding step in any round, nearest nodes and appended parents equal, states of every candidate that is not
                map-sensitive within 1e-5 at H = 32 (the metric's edge length) and within 1e-4 at the reference's H = 64
                (eight chained denoiser calls per candidate)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import rrt as ORRT
from oracle import sampler as OS
from tests.test_gpu_geometry import _scenario
from tests.test_gpu_round_precision import deviation, map_margin
from tests.util import REPO, golden

pytestmark = pytest.mark.gpu
BATCH, BUDGET, A, P = 64, 384, 8, 64


def make_onet():
    torch.manual_seed(0)
    onet = OD.init_noise_pred_net().eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in onet.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    return onet


def _sampler(onet, nz):
    def sampler(cand_idx, chunk, state, prev_action, has_prev, cond_goal, local_map):
        cv = OS.car_cond_vector(state, prev_action, has_prev, cond_goal)
        x1 = OS.flow_sample(onet, nz[cand_idx, chunk], OS.scale_local_map(local_map), cv, k_steps=1)
        return OS.unnormalize_actions(x1)
    return sampler


def _noise(tag, H):
    return torch.randn(BUDGET, H // A, P, 2, generator=torch.Generator().manual_seed(20260300 + len(tag)))


def oracle_plan(onet, tag, H, budget=BUDGET):
    """The FREE-RUNNING oracle plan of one scenario: a pure function of the seeds (cached under tests/golden/oracle_cache/)."""
    maze, start, goal = _scenario(golden("traces"), tag)
    pl = ORRT.OraclePlanner(maze, start, goal, _sampler(onet, _noise(tag, H).numpy()), edge_length=H, action_horizon=A)
    reached, path, actions = pl.plan(ORRT.RandomTape(42), budget, batch=BATCH)
    node = pl.goal_node if reached else pl.fallback_node()
    return dict(parents=np.array(pl.tree.parents), states=np.array(pl.tree.states), reached=bool(reached), iterations=pl.iterations,
                candidates=pl.candidates, sticky=bool(pl.sticky_triggered), node=-1 if node is None else int(node),
                path=path, actions=actions)


# rounds whose oracle expansion from the ENGINE's tree is recomputed live (it depends on the engine's own numbers, so it cannot
# be cached): every round at the metric's H = 32, the first three -- the third already hangs candidates under nodes of the
# first two -- at the reference's H = 64
PER_ROUND = {32: 99, 64: 3}


@pytest.fixture(scope="module")
def net_and_ctx():
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    onet = make_onet()
    ctx = Context(0)
    net = NoisePredNet(init=False)
    net.load_state_dict(onet.state_dict())
    net.bind(ctx, precision=_lib.PREC_F16X3, max_batch=BATCH)
    yield onet, ctx
    ctx.close()


# (scenario, edge length): the reference's car setting is prop_duration = [64] (8 chunks); BASELINE's metric is quoted at H = 32
@pytest.mark.parametrize("tag,H", [("race", 64), ("boxes", 64), ("rlarge2", 64), ("boxes", 32)])
def test_plan_with_the_denoiser_matches_the_oracle_plan(net_and_ctx, tag, H):
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    from tests.util import oracle_cache
    onet, ctx = net_and_ctx
    maze, start, goal = _scenario(golden("traces"), tag)
    noise = _noise(tag, H)
    sampler = _sampler(onet, noise.numpy())
    op, cached = oracle_cache(f"scenario_{tag}_H{H}", lambda: oracle_plan(onet, tag, H))
    if cached:                     # live probe: the plan's first round re-computed now
        live = oracle_plan(onet, tag, H, budget=BATCH)
        n = len(live["parents"])
        # (torch-CPU convolutions differ by ~1e-6 between host CPUs: the probe catches a stale cache, not rounding)
        assert np.array_equal(live["parents"], op["parents"][:n]) and np.abs(live["states"] - op["states"][:n]).max() < 2e-5, \
            "tests/golden/oracle_cache is stale: re-run make_oracle_cache.py"
    reached, path, actions = bool(op["reached"]), op["path"], op["actions"]

    eng = ExpansionEngine(ctx, maze, start, goal, edge_length=H, action_horizon=A, pred_horizon=P, batch=BATCH, capacity=4096)
    rt = ORRT.RandomTape(42)
    dev = ctx.device
    done = 0
    per_round = []
    while eng.goal_node is None and done < BUDGET:
        B = min(BATCH, BUDGET - done)
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        # the oracle on the ENGINE's tree as it is before this round
        n0 = eng.tree.n_nodes_host
        if len(per_round) >= PER_ROUND[H]:
            eng.expand_round(torch.as_tensor(s, device=dev), torch.as_tensor(c, device=dev), noise=noise[done:done + B].to(dev))
            done += B
            continue
        tf = ORRT.OraclePlanner(maze, start, goal, sampler, edge_length=H, action_horizon=A)
        t = tf.tree
        t.states = [r for r in eng.tree.state[:n0].cpu().numpy()]
        t.parents = [int(v) for v in eng.tree.parent[:n0].cpu().numpy()]
        t.last_action = [r for r in eng.tree.last_action[:n0].cpu().numpy()]
        t.has_prev = [bool(v) for v in eng.tree.has_prev[:n0].cpu().numpy()]
        t.num_visit = [int(v) for v in eng.tree.num_visit[:n0].cpu().numpy()]
        t.edge_states, t.edge_actions = [None] * n0, [None] * n0
        tf.candidates = done                                             # global candidate index = noise row
        tf.env_done_latched = bool(int(eng.tree.counters[2].item()))
        ref = tf.expand_round(s, c)
        run = np.arange(H // A)[None, :] < ref["chunks_run"][:, None]
        mm = np.full(run.shape, np.inf)
        mm[run] = map_margin(maze, ref["states"][:, :, 0][run])
        ref["map_margin"] = mm.min(axis=1)
        ref["tree_parents"], ref["tree_states"] = np.array(t.parents), np.array(t.states)
        eng.expand_round(torch.as_tensor(s, device=dev), torch.as_tensor(c, device=dev), noise=noise[done:done + B].to(dev))
        rb = eng.rb
        snap_r = eng.tree_snapshot()
        got = dict(status=rb.status[:B].cpu().numpy() & 0xFF, parent=rb.parent[:B].cpu().numpy(), end_state=rb.end_state[:B].cpu().numpy(),
                   states=rb.states[:B].cpu().numpy(), chunks_run=rb.chunks_run[:B].cpu().numpy(),
                   chunk_steps=rb.chunk_steps[:B].cpu().numpy(), tree_parents=snap_r["parents"], tree_states=snap_r["states"])
        per_round.append(deviation(got, ref))
        done += B
    snap = eng.tree_snapshot()
    ref_parents, ref_states = op["parents"], op["states"]
    print(tag, "nodes", len(ref_parents), "candidates", int(op["candidates"]), "iterations", int(op["iterations"]), "reached", reached)
    # ---- free-running: the discrete outputs are equal
    assert len(ref_parents) > 20                                      # a real tree, not a stump
    assert np.array_equal(snap["parents"], ref_parents)               # every accept decision of every round
    assert (eng.goal_node is not None) == reached
    assert int(snap["counters"][3]) == int(op["iterations"]) and int(snap["counters"][4]) == int(op["candidates"])
    assert int(snap["counters"][5]) == (1 if op["sticky"] else 0)
    node = eng.goal_node if reached else eng.fallback_node()
    assert node == (None if int(op["node"]) < 0 else int(op["node"]))
    p_eng, a_eng = eng.path_to(node)
    assert p_eng.shape == path.shape and a_eng.shape == actions.shape
    d_nodes = np.abs(snap["states"] - ref_states).max(axis=1)
    depth = np.zeros(len(ref_parents), dtype=int)
    for i in range(1, len(ref_parents)):
        depth[i] = depth[ref_parents[i]] + 1
    rec = {"nodes": int(len(ref_parents)), "candidates": int(op["candidates"]), "iterations": int(op["iterations"]), "reached": bool(reached),
           "max_depth": int(depth.max()), "free_running_node_state_deviation": {
               "median": float(np.median(d_nodes)), "p90": float(np.quantile(d_nodes, 0.9)), "p99": float(np.quantile(d_nodes, 0.99)),
               "max": float(d_nodes.max()), "share_within_1e-5": float((d_nodes < 1e-5).mean()),
               "median_by_depth": {int(k): float(np.median(d_nodes[depth == k])) for k in np.unique(depth)}},
           "path_deviation": float(np.abs(p_eng - path).max()),
           "per_round": [{k: v for k, v in d.items()} for d in per_round]}
    out = os.path.join(REPO, "gpurun_out", "scenario_parity.json")
    allr = {}
    if os.path.exists(out):
        with open(out) as f:
            allr = json.load(f)
    allr[f"{tag}_H{H}"] = rec
    with open(out, "w") as f:
        json.dump(allr, f, indent=1)
    # measured (profiles/r03_scenario_parity.json): the median deviation grows ~5x per tree level (3e-7, 2e-6, 7e-6, 4e-5,
    # 1e-4 at depth 1..5), 53 - 96 % of the nodes stay within 1e-5, single chains that crossed a map boundary reach 6e-2
    assert d_nodes.max() < 0.25 and np.median(d_nodes) < 1e-4, rec["free_running_node_state_deviation"]
    assert np.median(d_nodes[depth <= 2]) < 1e-5
    # ---- per round, from the engine's own tree: no flip anywhere; states of one expansion within the north star's 1e-5 at
    # its H = 32 -- and within 1e-4 at H = 64, where a candidate's deviation passes through eight chained denoiser calls
    # (measured maximum 3.9e-5, medians <= 1.6e-6)
    tol = 1e-5 if H <= 32 else 1e-4
    for r, dv in enumerate(per_round):
        assert dv["flips"] == 0 and dv["n_agree"] == dv["candidates"], (r, dv)
        assert dv["nn_parent_mismatches"] == 0 and dv["tree_parent_mismatches"] == 0, (r, dv)
        assert dv["max_abs_trajectory_state"] < tol and dv["median_abs_trajectory_state"] < 1e-5, (r, dv)
        assert dv["max_abs_state_map_sensitive"] < 5e-2, (r, dv)
