"""GPU: edge cases of the C-ABI -- empty and ragged batches, argument / call-order errors,
tree capacity overflow, maze limits."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle.tapes import ActionTape
from tests.util import load_maze

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def test_call_order_errors(ctx):
    from ditreeonlineplanner_amd._lib import DitreeError
    st = torch.zeros(4, 6, dtype=torch.float64, device="cuda")
    with pytest.raises(DitreeError, match="no maze"):
        ctx.local_map(st)
    with pytest.raises(DitreeError, match="no maze"):
        ctx.car_rollout(st, torch.zeros(4, 8, 2, dtype=torch.float64, device="cuda"), np.zeros(2))
    with pytest.raises(DitreeError, match="not loaded"):
        ctx.denoise(torch.zeros(4, 64, 2, device="cuda"), torch.zeros(4, 20, 20, device="cuda"), torch.zeros(4, 7, device="cuda"))
    with pytest.raises(DitreeError):
        ctx.upload_maze(np.zeros((300, 300), dtype=np.float32))          # beyond the LDS-staged limit
    with pytest.raises(TypeError):
        ctx.local_map(st.float())                                        # wrong dtype never reaches the kernels


def test_empty_batches_are_noops(ctx):
    maze = load_maze("boxes")
    ctx.upload_maze(maze)
    e6 = torch.zeros(0, 6, dtype=torch.float64, device="cuda")
    assert ctx.local_map(e6).shape == (0, 20, 20)
    status, states, aout, steps = ctx.car_rollout(e6, torch.zeros(0, 8, 2, dtype=torch.float64, device="cuda"), np.zeros(2))
    assert status.numel() == 0 and states.shape == (0, 9, 6)
    idx = ctx.nn_argmin(torch.zeros(0, 2, dtype=torch.float64, device="cuda"), torch.zeros(3, 2, dtype=torch.float64, device="cuda"))
    assert idx.numel() == 0
    d, e, h, v = ctx.lidar_scan(torch.zeros(0, 3, dtype=torch.float64, device="cuda"), dev(maze, torch.float32))
    assert d.shape == (0, 181)


@pytest.mark.parametrize("B", [1, 3, 63, 65, 257])
def test_ragged_batches_match_oracle(ctx, B):
    maze = load_maze("Race_Track")
    ctx.upload_maze(maze)
    rng = np.random.default_rng(B)
    free = np.argwhere(maze == 0)
    xy = G.cell_rowcol_to_xy(free[rng.integers(0, len(free), B)], maze) + rng.uniform(-0.4, 0.4, (B, 2))
    st = np.concatenate([xy, rng.uniform(-3, 3, (B, 1)), rng.uniform(0, 4, (B, 1)), rng.uniform(0, 1, (B, 1)),
                         rng.uniform(-0.4, 0.4, (B, 1))], axis=1)
    acts = np.stack([rng.uniform(-12, 12, (B, 8)), rng.uniform(-3, 3, (B, 8))], axis=2)
    goal = G.cell_rowcol_to_xy(np.array([1, 10]), maze)
    ref = G.rollout_chunk(st, acts, maze, goal, 8)
    state = dev(st)
    status, states, aout, steps = ctx.car_rollout(state, dev(acts), goal, A=8)
    assert np.array_equal(status.cpu().numpy() & 0xFF, ref["status"])
    assert np.abs(states.cpu().numpy() - ref["states"]).max() < 1e-9
    lm = ctx.local_map(dev(st)).cpu().numpy()
    assert np.array_equal(lm, G.create_local_map(maze.astype(np.float32), st[:, 0], st[:, 1], st[:, 2], 20, 0.2, 1.0,
                                                 (maze.shape[1] / 2, maze.shape[0] / 2)))
    nodes = rng.uniform(-3, 3, (B, 2))
    assert np.array_equal(ctx.nn_argmin(dev(st[:, :2].copy()), dev(nodes)).cpu().numpy(), G.nn_argmin(st[:, :2], nodes))


def test_capacity_overflow_is_flagged_not_faulting(ctx):
    from ditreeonlineplanner_amd.engine import CNT_NODES, CNT_OVERFLOW, ExpansionEngine
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    eng = ExpansionEngine(ctx, maze, start, goal, batch=64, capacity=24, emulate_sticky_done=False)
    from oracle import rrt as ORRT
    rt, at = ORRT.RandomTape(42), ActionTape(99)
    done = 0
    for _ in range(6):
        s, c = rt.draw_round(64, 20, 20, goal)
        acts = np.stack([at.actions(np.arange(done, done + 64), j) for j in range(eng.n_chunks)], axis=1)
        cnt = eng.expand_round(dev(s), dev(c), inject_actions=dev(acts))
        done += 64
    assert int(cnt[CNT_NODES]) == 24 and int(cnt[CNT_OVERFLOW]) == 1
    snap = eng.tree_snapshot()
    assert (snap["parents"][1:] >= 0).all() and (snap["parents"] < 24).all()


def test_denoise_argument_checks(ctx):
    from ditreeonlineplanner_amd._lib import DitreeError
    from ditreeonlineplanner_amd.model import NoisePredNet
    net = NoisePredNet(seed=1)
    net.bind(ctx, precision=0, max_batch=16)
    n = torch.zeros(32, 64, 2, device="cuda")
    with pytest.raises(DitreeError, match="exceeds"):
        ctx.denoise(n, torch.zeros(32, 20, 20, device="cuda"), torch.zeros(32, 7, device="cuda"))
    ctx.denoise_reserve(32, 0)                                            # growing the workspace is allowed
    out = ctx.denoise(n, torch.zeros(32, 20, 20, device="cuda"), torch.zeros(32, 7, device="cuda"), want_actions=False)
    assert torch.isfinite(out).all()
