"""GPU: the online driver steps (run_scenarios_with_lidar_DiTree.py:112-127,158-181,470-506) -- the fused
plan-following kernel and the reference-named host functions -- against tests/golden/online.npz, produced by
the reference's own functions / planner (make_golden.py gen_online) and against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import online as OO
from tests.util import golden

pytestmark = pytest.mark.gpu

CASES = ["visible", "free", "goal", "collision", "hidden", "track"]


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.mark.parametrize("tag", CASES)
def test_follow_plan_matches_reference_loop(ctx, tag):
    g = golden("online")
    k = lambda n: g[f"online_{tag}_{n}"]
    ctx.upload_maze(k("maze").astype(np.float32))
    st = dev(k("start"))
    known, scanned = dev(k("known0"), torch.float32), dev(k("scanned0"), torch.float32)
    ex, event, nxt, obstacle = ctx.follow_plan(st, dev(k("actions"), torch.float32), 0, dev(k("path")[:, :2], torch.float32),
                                               known, dev(k("true"), torch.float32), scanned, k("goal_xy"), 0.02, 0.2)
    assert [event, nxt, obstacle] == [int(v) for v in k("result")]
    assert ex.shape == k("executed").shape and (ex.numel() == 0 or np.abs(ex.cpu().numpy() - k("executed")).max() < 1e-9)
    assert np.abs(st.cpu().numpy() - k("state")).max() < 1e-9
    assert np.array_equal(known.cpu().numpy(), k("known1")) and np.array_equal(scanned.cpu().numpy(), k("scanned1"))
    # the ctx's own maze copy was refreshed: collisions against the new known maze without an upload
    probe = np.argwhere(k("known1") != k("maze"))
    if len(probe):
        xy = G.cell_rowcol_to_xy(probe[0], k("maze"))
        s = dev(np.array([[xy[0], xy[1], 0.0, 0, 0, 0]]))
        status, _, _, _ = ctx.car_rollout(s, dev(np.zeros((1, 1, 2))), np.array([1e6, 1e6]), A=1)
        assert int(status.item()) & 0xFF == 2


def test_follow_plan_resumes_mid_plan(ctx):
    """Starting at action k from the state reached after k actions gives the tail of the full run."""
    g = golden("online")
    k = lambda n: g[f"online_hidden_{n}"]
    cut = 300
    ctx.upload_maze(k("maze").astype(np.float32))
    known, scanned = k("known0").copy(), k("scanned0").copy()
    o = OO.follow_plan(k("executed")[cut - 1], k("actions"), cut, k("path"), known, k("true"), scanned, k("goal_xy"))
    st = dev(k("executed")[cut - 1])
    dk, ds = dev(k("known0"), torch.float32), dev(k("scanned0"), torch.float32)
    ex, event, nxt, obstacle = ctx.follow_plan(st, dev(k("actions"), torch.float32), cut, dev(k("path")[:, :2], torch.float32),
                                               dk, dev(k("true"), torch.float32), ds, k("goal_xy"), 0.02, 0.2)
    assert (event, nxt, obstacle) == (o["event"], o["action_idx"], o["obstacle_idx"])
    assert np.abs(ex.cpu().numpy() - o["executed"]).max() < 1e-9
    assert np.array_equal(dk.cpu().numpy(), known) and np.array_equal(ds.cpu().numpy(), scanned)
    # nothing left to execute: no-op
    ex, event, nxt, obstacle = ctx.follow_plan(st, dev(k("actions"), torch.float32), len(k("actions")),
                                               dev(k("path")[:, :2], torch.float32), dk, dev(k("true"), torch.float32), ds,
                                               k("goal_xy"), 0.02, 0.2)
    assert (event, nxt, obstacle, ex.shape[0]) == (0, len(k("actions")), -1, 0)


class _Planner:
    """The attributes the driver functions touch (planner.env, planner.update_maze, planner.ctx)."""

    def __init__(self, ctx, maze):
        from ditreeonlineplanner_amd.car_env import CarEnv
        self.ctx = ctx
        self.env = CarEnv(maze_map=maze.copy(), collision_checking=False, ctx=ctx)
        self.maze = maze.copy()
        self.updates = 0

    def update_maze(self, m):
        self.maze = np.float32(m)
        self.env.maze_map = m
        self.updates += 1

    adopt_maze = update_maze


def _reset(pl, start, goal, maze):
    env = pl.env
    env.reset(options={"reset_cell": env.cell_xy_to_rowcol(start[:2]), "reset_deg": np.rad2deg(start[2]),
                       "goal_cell": env.cell_xy_to_rowcol(goal[:2])})


@pytest.mark.parametrize("tag", ["visible", "hidden"])
def test_driver_functions_match_reference(ctx, tag):
    from ditreeonlineplanner_amd import online
    g = golden("online")
    k = lambda n: g[f"online_{tag}_{n}"]
    maze = k("maze")
    pl = _Planner(ctx, maze)
    goal = np.array([*k("goal_xy"), 0, 0, 0, 0])
    _reset(pl, k("start"), goal, maze)
    known, scanned = maze.copy(), maze.copy()
    online.scan_and_update_maze(pl, known, k("true"), scanned)            # the initial scan (float32 env state)
    assert np.array_equal(known, k("known0")) and np.array_equal(scanned, k("scanned0")) and pl.updates == 1
    # the reference's own loop shape on the facade functions, a scan after every 11th step
    pl.env.set_state(k("executed")[10])
    online.scan_and_update_maze(pl, known, k("true"), scanned)
    ok, os_ = k("known0").copy(), k("scanned0").copy()
    OO.scan_and_update_maze(k("executed")[10], ok, k("true"), os_)
    assert np.array_equal(known, ok) and np.array_equal(scanned, os_)
    assert online.check_no_obstacles_in_path(pl, scanned, k("path")) == OO.check_no_obstacles_in_path(os_, k("path"))
    # fused launch through the facade
    known, scanned = k("known0").copy(), k("scanned0").copy()
    state, nxt, executed, event, obstacle = online.follow_plan(pl, k("start"), k("actions"), 0, k("path"), known,
                                                               k("true"), scanned)
    assert [event, nxt, obstacle] == [int(v) for v in k("result")]
    assert np.abs(executed - k("executed")).max() < 1e-9 and np.abs(state - k("state")).max() < 1e-9
    assert np.array_equal(known, k("known1")) and np.array_equal(scanned, k("scanned1"))
    assert np.array_equal(pl.maze, k("known1").astype(np.float32)) and np.abs(pl.env.state - k("state")).max() < 1e-9


def test_path_check_known_answers(ctx):
    from ditreeonlineplanner_amd import online
    g = golden("online")
    maze, path = g["online_free_maze"], g["online_free_path"]
    pl = _Planner(ctx, maze)
    for marks, exp in zip(g["online_check_marks"], g["online_check_expected"]):
        sc = maze.copy()
        sc[marks[:, 0], marks[:, 1]] = 1
        assert online.check_no_obstacles_in_path(pl, sc, path) == int(exp)
