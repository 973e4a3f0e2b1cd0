"""CPU: the ant glue of the oracle (oracle/ant.py) against the goldens written from the reference's own functions and from
the reference's RRT_Planner(env_id='antmaze') on a stand-in env (tests/golden/make_golden.py antglue -> ant.npz)."""
import numpy as np
import pytest

from oracle import ant as OA
from oracle import rrt as ORRT
from tests.util import golden, load_maze

TRACES = ("tape_boxes", "tape_xlarge", "tape_val7", "model_boxes", "model_val7")


def trace_setup(tag):
    """(golden dict view, oracle planner, action tape, step_fn, obs tape or None) of one golden ant trace."""
    g = golden("ant")
    pre = f"anttrace_{tag}_"
    seed, cands, iters, reached, goal_node, is_model = (int(v) for v in g[pre + "meta"])
    maze = g[pre + "maze"]
    atape = OA.AntActionTape(seed, 16)
    otape = None
    if is_model:
        step_fn = lambda cand, chunk, i, cur, act: OA.ant_model_step(cur, act)     # noqa: E731
    else:
        otape = OA.AntObsTape(seed + 1, maze, 4.0, 24, 2, desired_xy=g[pre + "desired"], goal_every=29, step=0.12)
        step_fn = otape.step_fn()
    pl = OA.OracleAntPlanner(maze, g[pre + "start"], g[pre + "goal"], g[pre + "desired"], atape.sampler(), step_fn)
    return g, pre, pl, atape, otape, dict(seed=seed, candidates=cands, iterations=iters, reached=bool(reached), goal_node=goal_node,
                                          is_model=bool(is_model), maze=maze)


def test_ant_collision_matches_reference():
    g = golden("ant")
    for name in ("Race_Track", "boxes", "random_huge", "narrow_short"):
        st = g[f"antcol_{name}_states"]
        got = OA.is_colliding_ant(st, load_maze(name), 1.2, 4.0)
        assert np.array_equal(got, g[f"antcol_{name}_expected"]), name
        assert 0.2 < got.mean() < 0.95
    got = OA.is_colliding_ant(g["antcol_unit_states"], load_maze("boxes"), 0.3, 1.0)
    assert np.array_equal(got, g["antcol_unit_expected"])


@pytest.mark.parametrize("tag", TRACES)
def test_oracle_ant_planner_reproduces_the_reference_trace(tag):
    g, pre, pl, atape, otape, m = trace_setup(tag)
    calls = []
    inner = pl.sampler

    def rec(cand_idx, chunk, hist, prev_a, has_prev, cond_goal, lm):
        for k, c in enumerate(cand_idx):
            calls.append((int(c), chunk, hist[k].copy(), prev_a[k].copy(), bool(has_prev[k]), cond_goal[k].copy(), lm[k].copy()))
        return inner(cand_idx, chunk, hist, prev_a, has_prev, cond_goal, lm)
    pl.sampler = rec
    reached, path, actions = pl.plan(ORRT.RandomTape(42), m["candidates"], batch=1)
    assert reached == m["reached"] and pl.iterations == m["iterations"] and pl.candidates == m["candidates"]
    assert np.array_equal(np.array(pl.parents), g[pre + "parents"])
    assert np.array_equal(np.array(pl.states), g[pre + "states"])
    assert np.array_equal(path, g[pre + "path"]) and np.array_equal(actions, g[pre + "actions"])
    if reached:
        assert pl.goal_node == m["goal_node"]
    # what the reference planner handed its sampler (RRT.py:146-147,168-171,186-190): history rows, previous action, goal, map
    key = g[pre + "call_key"]
    lm_all = np.unpackbits(g[pre + "call_lmap"])[: len(key) * 256].reshape(len(key), 16, 16)
    for k in range(len(key)):
        c, j, hist, pa, hp, goal, lm = calls[k]
        assert (c, j, len(hist), int(hp)) == tuple(key[k])
        n = len(hist)
        assert np.array_equal(hist, g[pre + "call_hist"][k][3 - n:])
        assert (not hp) or np.array_equal(pa, g[pre + "call_prev"][k])
        assert np.array_equal(goal, g[pre + "call_goal"][k]) and np.array_equal(lm.astype(np.uint8), lm_all[k])


def test_oracle_rounds_do_not_depend_on_the_sampler_grouping():
    """A round of B candidates = the same candidates' edges as B = 1 rounds would give against the same snapshot: statuses of
    the first round agree between batch sizes (the tree snapshot is the root for both)."""
    g, pre, pl1, atape, otape, m = trace_setup("tape_boxes")
    _, _, pl8, _, _, _ = trace_setup("tape_boxes")
    t1, t8 = ORRT.RandomTape(42), ORRT.RandomTape(42)
    s = np.zeros((8, 29))
    c = np.zeros((8, 2))
    for i in range(8):
        s[i], c[i] = OA.draw_candidate_ant(t8, 20, 20, 4.0, pl8.goal_state)
    r8 = pl8.expand_round(s, c)
    r1 = pl1.expand_round(s[:1], c[:1])
    assert r8["status"][0] == r1["status"][0] and np.array_equal(r8["states"][0], r1["states"][0])


def test_ant_model_is_finite_and_keeps_the_quaternion_unit():
    rng = np.random.default_rng(0)
    s = np.zeros((64, 29))
    s[:, 2], s[:, 3] = 0.75, 1.0
    s[:, 7:15] = np.tile([0.0, OA.AntModel.ank_rest], 4)
    for k in range(60):
        s = OA.ant_model_step(s, rng.uniform(-1.5, 1.5, (64, 8)))
    assert np.isfinite(s).all() and np.abs(np.linalg.norm(s[:, 3:7], axis=1) - 1).max() < 1e-12
    assert np.abs(s[:, :2]).max() > 1e-3 and (OA.body_z_up(s[:, 3:7]) > 0.5).all()
    assert np.allclose(OA.AntModel.as_vector()[:3], [0.01, 5, 60.0])
