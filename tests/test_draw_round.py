"""CPU: the bulk draw of a round (planners/_draw.py) equals the reference-order loop element for element and leaves `random`
and `np.random` in the same state -- every run_type, with and without a remaining reference path, car and ant."""
import random
import types

import numpy as np
import pytest

from ditreeonlineplanner_amd.planners import RRT as F
from ditreeonlineplanner_amd.planners._draw import draw_round_bulk
from ditreeonlineplanner_amd.planners.base_planner import BasePlanner, Node
from tests.util import load_maze


class _Env:
    def __init__(self, maze, prob_map):
        self.maze_map = maze
        self.prob_map = prob_map

    def cell_rowcol_to_xy(self, rc):
        rc = np.asarray(rc)
        H, W = self.maze_map.shape
        return np.array([(rc[1] + 0.5) * 1.0 - W / 2, H / 2 - (rc[0] + 0.5) * 1.0])


def fake_planner(run_type, env_id="carmaze", maze_name="boxes"):
    """The attributes the draw reads, on a plain object: the facade's own methods run on it unbound (no GPU context)."""
    maze = load_maze(maze_name)
    H, W = maze.shape
    rng = np.random.default_rng(5)
    pm = rng.random(maze.shape) * (maze == 0)
    pm /= pm.sum()
    p = types.SimpleNamespace()
    p.env_id, p.run_type = env_id, run_type
    p.env = _Env(maze, pm)
    S = 29 if "ant" in env_id else 6
    p.start_node = Node(np.zeros(S))
    p.goal_state = np.zeros(S)
    p.goal_state[:2] = [7.5, 7.5]
    p.goal_sample_rate, p.goal_conditioning_bias = 0.15, 0.85
    p.map_width, p.map_length, p.max_v, p.s_global = W, H, 5, 4.0 if "ant" in env_id else 1.0
    p.random_node_sample = types.MethodType(BasePlanner.random_node_sample, p)
    p.sample_row_col_from_probability_map = types.MethodType(BasePlanner.sample_row_col_from_probability_map, p)
    p._draw_round_loop = types.MethodType(F.RRT_Planner._draw_round_loop, p)
    return p


def states():
    return random.getstate(), np.random.get_state()


def same_state(a, b):
    return a[0] == b[0] and a[1][0] == b[1][0] and np.array_equal(a[1][1], b[1][1]) and a[1][2:] == b[1][2:]


@pytest.mark.parametrize("run_type", [0, 1, 2, 3])
@pytest.mark.parametrize("with_path", [False, True])
def test_bulk_draw_equals_the_loop(run_type, with_path):
    if run_type == 0 and with_path:
        pytest.skip("run_type 0 never has a remaining path")
    p = fake_planner(run_type)
    for trial, (B, n_path) in enumerate([(1, 1), (7, 3), (64, 17), (1000, 100), (3000, 1), (513, 1025)]):
        path = None
        if with_path:
            path = np.random.default_rng(trial).uniform(-9, 9, (n_path, 2)).astype(np.float32)
        random.seed(100 + trial)
        np.random.seed(200 + trial)
        s_ref, c_ref = p._draw_round_loop(B, path)
        after_ref = states()
        nxt_ref = (random.random(), np.random.random_sample(), np.random.randint(0, 7))
        random.seed(100 + trial)
        np.random.seed(200 + trial)
        out = draw_round_bulk(p, B, path)
        assert out is not None
        after = states()
        nxt = (random.random(), np.random.random_sample(), np.random.randint(0, 7))
        assert np.array_equal(out[0], s_ref) and np.array_equal(out[1], c_ref), (run_type, with_path, B)
        assert same_state(after, after_ref) and nxt == nxt_ref
        assert out[0].dtype == np.float64 and out[0].shape == (B, 6)


def test_bulk_draw_ant():
    p = fake_planner(0, env_id="antmaze")
    for B in (1, 33, 2000):
        random.seed(B)
        np.random.seed(B + 1)
        s_ref, c_ref = p._draw_round_loop(B, None)
        after_ref = states()
        random.seed(B)
        np.random.seed(B + 1)
        s, c = draw_round_bulk(p, B, None)
        assert np.array_equal(s, s_ref) and np.array_equal(c, c_ref) and same_state(states(), after_ref)
        assert s.shape == (B, 29) and not s[:, 2:].any() and np.abs(s[:, :2]).max() <= 40.0


def test_bulk_draw_is_fast():
    import time
    p = fake_planner(0)
    random.seed(1)
    np.random.seed(1)
    t_bulk = t_loop = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        draw_round_bulk(p, 8192, None)
        t_bulk = min(t_bulk, time.perf_counter() - t0)
    for _ in range(2):
        t0 = time.perf_counter()
        p._draw_round_loop(8192, None)
        t_loop = min(t_loop, time.perf_counter() - t0)
    assert t_bulk < t_loop / 8, (t_bulk, t_loop)
