"""Scenario-level parity WITH THE DENOISER IN THE LOOP (the north star's wording: "outputs match the CPU reference per
scenario -- reached-goal flag and tree parent indices bit-exact, propagated states within 1e-5").

The golden traces pin the planner mechanics with tape actions; tests/test_gpu_round_precision.py pins single rounds with the
network.  Here whole PLANS run on both sides: scenarios of experiments/test_scenarios_car.csv (the rows the golden traces
use), `prop_duration = [64]` (8 chunks of 8 steps, the reference's car setting), rounds of 64 candidates drawn in the
reference's RNG order (seed 42), a candidate budget instead of the wall clock, the same seeded weights and noise -- engine
(f16x3, the default instantiation) vs oracle planner (numpy f64 geometry + torch-CPU fp32 denoiser).  A single flipped flag
anywhere would change every later parent index, so equality of the final trees is a statement about every candidate of
every round."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import rrt as ORRT
from oracle import sampler as OS
from tests.test_gpu_geometry import _scenario
from tests.util import golden

pytestmark = pytest.mark.gpu
BATCH, BUDGET, H, A, P = 64, 384, 64, 8, 64


@pytest.fixture(scope="module")
def net_and_ctx():
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    onet = OD.init_noise_pred_net().eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in onet.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    ctx = Context(0)
    net = NoisePredNet()
    net.load_state_dict(onet.state_dict())
    net.bind(ctx, precision=_lib.PREC_F16X3, max_batch=BATCH)
    yield onet, ctx
    ctx.close()


@pytest.mark.parametrize("tag", ["race", "boxes", "rlarge2"])
def test_plan_with_the_denoiser_matches_the_oracle_plan(net_and_ctx, tag):
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    onet, ctx = net_and_ctx
    maze, start, goal = _scenario(golden("traces"), tag)
    gen = torch.Generator().manual_seed(20260300 + len(tag))
    noise = torch.randn(BUDGET, H // A, P, 2, generator=gen)
    nz = noise.numpy()

    def sampler(cand_idx, chunk, state, prev_action, has_prev, cond_goal, local_map):
        cv = OS.car_cond_vector(state, prev_action, has_prev, cond_goal)
        x1 = OS.flow_sample(onet, nz[cand_idx, chunk], OS.scale_local_map(local_map), cv, k_steps=1)
        return OS.unnormalize_actions(x1)

    pl = ORRT.OraclePlanner(maze, start, goal, sampler, edge_length=H, action_horizon=A)
    reached, path, actions = pl.plan(ORRT.RandomTape(42), BUDGET, batch=BATCH)

    eng = ExpansionEngine(ctx, maze, start, goal, edge_length=H, action_horizon=A, pred_horizon=P, batch=BATCH, capacity=4096)
    rt = ORRT.RandomTape(42)
    dev = ctx.device
    done = 0
    while eng.goal_node is None and done < BUDGET:
        B = min(BATCH, BUDGET - done)
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        eng.expand_round(torch.as_tensor(s, device=dev), torch.as_tensor(c, device=dev), noise=noise[done:done + B].to(dev))
        done += B
    snap = eng.tree_snapshot()
    ref_parents, ref_states = np.array(pl.tree.parents), np.array(pl.tree.states)
    print(tag, "nodes", len(ref_parents), "candidates", pl.candidates, "iterations", pl.iterations, "reached", reached)
    assert len(ref_parents) > 20                                      # a real tree, not a stump
    assert np.array_equal(snap["parents"], ref_parents)               # every accept decision of every round
    assert np.abs(snap["states"] - ref_states).max() < 1e-5
    assert (eng.goal_node is not None) == reached
    assert int(snap["counters"][3]) == pl.iterations and int(snap["counters"][4]) == pl.candidates
    assert int(snap["counters"][5]) == (1 if pl.sticky_triggered else 0)
    node = eng.goal_node if reached else eng.fallback_node()
    assert node == (pl.goal_node if reached else pl.fallback_node())
    p_eng, a_eng = eng.path_to(node)
    assert p_eng.shape == path.shape and np.abs(p_eng - path).max() < 1e-5
    assert a_eng.shape == actions.shape and np.abs(a_eng - actions).max() < 1e-4
