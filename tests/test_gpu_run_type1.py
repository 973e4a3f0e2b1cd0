"""GPU: run_type 1 ("Original+Ref") -- obstacle-ahead flags, reference-path sampling and the
furthest-along-path fallback (planners/RRT.py:61-111,134-140,153-156,202-254) against the trace the
reference planner produced (tests/golden/traces.npz, rt1_* keys) and against the CPU oracle."""
import random

import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import rrt as ORRT
from oracle.tapes import ActionTape
from tests.util import golden, load_maze

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def test_obstacle_ahead_bit_exact(ctx):
    g = golden("traces")
    maze = load_maze("boxes")
    ctx.upload_maze(maze.astype(np.float32))
    poses = g["rt1_ahead_poses"]
    st = np.concatenate([poses, np.zeros((len(poses), 3))], axis=1)
    got = ctx.obstacle_ahead(dev(st)).cpu().numpy().astype(bool)
    assert np.array_equal(got, g["rt1_ahead_expected"])
    assert 0.2 < got.mean() < 0.8
    # poses outside the map are clipped onto the border cells like the reference's np.clip
    rng = np.random.default_rng(3)
    far = np.concatenate([rng.uniform(-14, 14, (256, 2)), rng.uniform(-np.pi, np.pi, (256, 1)), np.zeros((256, 3))], axis=1)
    assert np.array_equal(ctx.obstacle_ahead(dev(far)).cpu().numpy().astype(bool), G.check_obstacle_ahead(far, maze))
    assert ctx.obstacle_ahead(dev(np.zeros((0, 6)))).numel() == 0


def _run(ctx, maze, start, goal, seed, budget, batch, init_main_path=None, remain=None, run_type=1, prob_map=None):
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    eng = ExpansionEngine(ctx, maze, start, goal, batch=batch, capacity=2048, run_type=run_type, early_exit=True)
    eng.init_main_path = init_main_path
    rt, at = ORRT.RandomTape(42), ActionTape(seed)
    done = 0
    while eng.goal_node is None and done < budget:
        B = min(batch, budget - done)
        s, c = np.zeros((B, 6)), np.zeros((B, 2))
        for i in range(B):
            s[i], c[i] = rt.draw_candidate_ref(remain, maze.shape[1], maze.shape[0], goal, prob_map=prob_map)
        acts = np.stack([at.actions(np.arange(done, done + B), j) for j in range(eng.n_chunks)], axis=1)
        eng.expand_round(dev(s), dev(c), inject_actions=dev(acts))
        done += B
    return eng


def _check_tree(eng, parents, states):
    snap = eng.tree_snapshot()
    assert np.array_equal(snap["parents"], parents)
    assert np.abs(snap["states"] - states).max() < 1e-9
    return snap


def test_engine_reproduces_reference_run_type1_trace(ctx):
    """Both stages of the golden trace with B = 1 rounds: the tree (parents, states), the returned path and
    actions equal the reference planner's; stage 2 samples along the remaining reference path and ends in the
    furthest-along-path fallback or the goal, whichever the reference took."""
    g = golden("traces")
    maze = load_maze("boxes")
    start, goal, seed = g["rt1_start"], g["rt1_goal"], int(g["rt1_seed"])
    n1, n2 = [int(v) for v in g["rt1_budgets"]]
    eng = _run(ctx, maze, start, goal, seed, n1, 1)
    _check_tree(eng, g["rt1_parents1"], g["rt1_states1"])
    n = eng.tree.n_nodes_host
    flags = eng.tree.obstacle_ahead[1:n].cpu().numpy().astype(bool)
    assert np.array_equal(flags, G.check_obstacle_ahead(g["rt1_states1"][1:], maze))
    node = eng.goal_node if eng.goal_node is not None else eng.fallback_node()
    path1, act1 = eng.path_to(node)
    assert path1.shape == g["rt1_path1"].shape and np.abs(path1 - g["rt1_path1"]).max() < 1e-5
    assert np.array_equal(act1, g["rt1_actions1"])

    maze2 = g["rt1_maze2"]
    ref_path = g["rt1_path1"]
    opl = ORRT.OraclePlanner(maze2, start, goal, ActionTape(seed).sampler(), run_type=1, init_main_path=ref_path)
    remain = opl.remaining_reference_path()
    assert 0 < len(remain) < len(ref_path)
    eng2 = _run(ctx, maze2, start, goal, seed, n2, 1, init_main_path=ref_path, remain=remain)
    _check_tree(eng2, g["rt1_parents2"], g["rt1_states2"])
    node2 = eng2.goal_node if eng2.goal_node is not None else eng2.fallback_node()
    if bool(g["rt1_has_path2"]):
        path2, act2 = eng2.path_to(node2)
        assert path2.shape == g["rt1_path2"].shape and np.abs(path2 - g["rt1_path2"]).max() < 1e-5
        assert np.array_equal(act2, g["rt1_actions2"])
    else:
        assert node2 is None


@pytest.mark.parametrize("batch", [16, 128])
def test_run_type1_rounds_vs_oracle(ctx, batch):
    """Wide rounds: flags per node, both fallback rules and the all-flagged 'no plan' outcome vs the oracle."""
    g = golden("traces")
    maze2, start, goal = g["rt1_maze2"], g["rt1_start"], g["rt1_goal"]
    ref_path = g["rt1_path1"]
    budget = batch * 4
    for with_path in (False, True):
        opl = ORRT.OraclePlanner(maze2, start, goal, ActionTape(5).sampler(), run_type=1,
                                 init_main_path=ref_path if with_path else None)
        remain = opl.remaining_reference_path() if with_path else None
        reached, opath, oact = opl.plan(ORRT.RandomTape(42), budget, batch=batch)
        eng = _run(ctx, maze2, start, goal, 5, budget, batch, init_main_path=ref_path if with_path else None,
                   remain=remain)
        _check_tree(eng, np.array(opl.tree.parents), np.array(opl.tree.states))
        n = eng.tree.n_nodes_host
        assert np.array_equal(eng.tree.obstacle_ahead[1:n].cpu().numpy().astype(bool), np.array(opl.obstacle_ahead, dtype=bool))
        assert (eng.goal_node is not None) == reached
        node = eng.goal_node if reached else eng.fallback_node()
        if opath is None:
            assert node is None
        else:
            path, act = eng.path_to(node)
            assert path.shape == opath.shape and np.abs(path - opath).max() < 1e-5 and np.array_equal(act, oact)


@pytest.mark.parametrize("rt", [2, 3])
def test_engine_reproduces_reference_run_type23_trace(ctx, rt):
    """run_type 2 / 3: positions drawn from the sampling-probability map (host RNG order), everything else as
    run_type 1 -- tree, path and actions equal the reference planner's."""
    g = golden("traces")
    maze = load_maze("boxes")
    start, goal = g["rt1_start"], g["rt1_goal"]
    eng = _run(ctx, maze, start, goal, int(g[f"rt{rt}_seed"]), int(g[f"rt{rt}_budget"]), 1, run_type=rt,
               prob_map=g[f"rt{rt}_prob_map"])
    _check_tree(eng, g[f"rt{rt}_parents"], g[f"rt{rt}_states"])
    node = eng.goal_node if eng.goal_node is not None else eng.fallback_node()
    path, act = eng.path_to(node)
    assert path.shape == g[f"rt{rt}_path"].shape and np.abs(path - g[f"rt{rt}_path"]).max() < 1e-5
    assert np.array_equal(act, g[f"rt{rt}_actions"])


def test_fallback_none_when_every_node_has_an_obstacle_ahead(ctx):
    """RRT.py:227-232: np.all(has_obstacle_ahead) -> (None, None); also true for a tree holding only the start."""
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    g = golden("traces")
    maze, start, goal = load_maze("boxes"), g["rt1_start"], g["rt1_goal"]
    eng = ExpansionEngine(ctx, maze, start, goal, batch=8, capacity=64, run_type=1)
    assert eng.fallback_node() is None
    eng = _run(ctx, maze, start, goal, 77, 40, 8)
    n = eng.tree.n_nodes_host
    assert n > 2
    eng.tree.obstacle_ahead[1:n] = 1
    assert eng.fallback_node() is None
    eng.tree.obstacle_ahead[n - 1] = 0
    assert eng.fallback_node() == n - 1


def test_planner_facade_run_type1(ctx):
    """RRT_Planner(run_type=1): extract_path_after_obstacle and the sampling order equal the oracle's; plan()
    runs on the denoiser with a reference path and reports like the reference."""
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler
    from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net
    g = golden("traces")
    maze2, start, goal, ref_path = g["rt1_maze2"], g["rt1_start"], g["rt1_goal"], g["rt1_path1"]
    torch.manual_seed(0)
    net = init_noise_pred_net(input_dim=2, action_dim=2, obs_dim=3, obs_history=1, action_history=1,
                              goal_conditioned=True, goal_dim=2, local_map_conditioned=True,
                              local_map_encoder="resnet", local_map_embedding_dim=400, local_map_size=20,
                              down_dims=[512, 1024, 2048])
    smp = DiffusionSampler(net, None, "carmaze", policy="flow_matching", pred_horizon=64, action_dim=2,
                           prediction_type="actions", obs_history=1, action_history=1, goal_conditioned=True,
                           num_diffusion_iters=1, local_map_size=20).eval()
    env = CarEnv(maze_map=load_maze("boxes"), collision_checking=False, run_type=1)
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=smp, prediction_type="actions",
                     action_horizon=8, local_map_size=20, local_map_scale=0.2, global_map_scale=1.0,
                     goal_conditioning_bias=0.85, prop_duration=[64], time_budget=120, max_iter=300, verbose=False,
                     batch=32, max_candidates=96, run_type=1)
    pl.update_maze(maze2)
    pl.init_main_path = ref_path.copy()
    pl.reset(start_state=start, goal_state=goal)
    opl = ORRT.OraclePlanner(maze2, start, goal, ActionTape(1).sampler(), run_type=1, init_main_path=ref_path)
    remain = pl.extract_path_after_obstacle()
    assert np.array_equal(remain, opl.remaining_reference_path())
    random.seed(42)
    np.random.seed(42)
    s, c = pl.draw_round(64, remain)
    rt = ORRT.RandomTape(42)
    for i in range(64):
        es, ec = rt.draw_candidate_ref(remain, maze2.shape[1], maze2.shape[0], goal)
        assert np.array_equal(s[i, :2], es[:2]) and np.array_equal(c[i], ec)
    assert pl.check_obstacle_ahead(start) == bool(G.check_obstacle_ahead(start[None], maze2)[0])
    random.seed(1)
    np.random.seed(1)
    torch.manual_seed(1)
    path, actions = pl.plan()
    r = pl.results
    assert r["iterations"] > 0 and r["number_of_nodes"] >= 1
    n = pl._engine.tree.n_nodes_host
    flags = pl._engine.tree.obstacle_ahead[1:n].cpu().numpy().astype(bool)
    states = pl._engine.tree.state[1:n].cpu().numpy()
    assert np.array_equal(flags, G.check_obstacle_ahead(states, maze2)) if n > 1 else True
    if path is None:
        assert n == 1 or flags.all()
    else:
        assert path.dtype == np.float32 and path.shape[1] == 6 and np.allclose(path[0], start.astype(np.float32))
    # run_type 3: the facade's sampling order (map draw, then the uniform draws) equals the reference's
    env3 = CarEnv(maze_map=load_maze("boxes"), collision_checking=False, run_type=3)
    pl3 = RRT_Planner(start, goal, env_id="carmaze", environment=env3, sampler=smp, prediction_type="actions",
                      action_horizon=8, local_map_size=20, local_map_scale=0.2, global_map_scale=1.0,
                      prop_duration=[64], time_budget=120, batch=32, max_candidates=64, run_type=3)
    pl3.reset()
    env3.update_prob_map_by_loc()
    assert np.array_equal(env3.prob_map, g["rt3_prob_map"])
    random.seed(42)
    np.random.seed(42)
    s3, c3 = pl3.draw_round(48, None)
    rt3 = ORRT.RandomTape(42)
    for i in range(48):
        es, ec = rt3.draw_candidate_ref(None, 20, 20, goal, prob_map=g["rt3_prob_map"])
        assert np.array_equal(s3[i], es) and np.array_equal(c3[i], ec)
    before = env3.prob_map.copy()
    path3, _ = pl3.plan()
    assert np.array_equal(env3.prob_map, before) and pl3.results["iterations"] > 0


def test_extract_path_after_obstacle_against_the_reference_function(ctx):
    """ditree_path_after_obstacle against goldens written by the reference's OWN RRT_Planner.extract_path_after_obstacle
    (planners/RRT.py:83-111; tests/golden/make_golden.py pao): float32 paths, float64 and float32 env states -- numpy forms the
    nearest-point distances in float64 when env.state is float64 (the f32 path is promoted) and in float32 right after
    env.reset; the kernel does the same.  Checked: the nearest index itself and the returned remainder of the path."""
    import ctypes as C
    from ditreeonlineplanner_amd._lib import check, lib
    g = golden("traces")
    n = int(g["pao_n"])
    assert n >= 80
    seen64 = seen32 = 0
    for i in range(n):
        path, st, f32 = g[f"pao_{i}_path"], g[f"pao_{i}_state"], bool(g[f"pao_{i}_f32"])
        mz = np.unpackbits(g[f"pao_{i}_maze"])[:400].reshape(20, 20).astype(np.float32)
        exp = g[f"pao_{i}_expected"]
        ctx.upload_maze(mz)
        p_dev = dev(path)
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        cur = st.astype(np.float32).astype(np.float64) if f32 else st
        check(ctx._h, lib().ditree_path_after_obstacle(ctx._h, p_dev.data_ptr(), 2, len(path), (C.c_double * 2)(*cur), int(f32),
                                                        out.data_ptr(), ctx.stream), "path_after_obstacle")
        c, k = (int(v) for v in out.cpu().numpy())
        c_ref = int(np.argmin(np.linalg.norm((st.astype(np.float32) if f32 else st) - path, axis=1)))
        assert c == c_ref, (i, f32, c, c_ref)
        got = path[c:][k:]
        assert np.array_equal(got, exp) and np.array_equal(ORRT.path_after_obstacle(path, st.astype(np.float32) if f32 else st, mz), exp)
        seen64 += not f32
        seen32 += f32
    assert seen64 >= 40 and seen32 >= 40
