"""GPU: the reference-shaped Python surface (CarEnv, Lidar2DSim, DiffusionSampler, RRT_Planner)
driving the HIP engine; written like the tests the reference would have for these classes."""
import random

import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import geometry as G
from oracle import rrt as ORRT
from oracle import sampler as OS
from tests.util import golden, load_maze

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net_pair():
    from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net
    torch.manual_seed(0)
    onet = OD.init_noise_pred_net().eval()
    net = init_noise_pred_net(input_dim=2, action_dim=2, obs_dim=3, obs_history=1, action_history=1,
                              goal_conditioned=True, goal_dim=2, local_map_conditioned=True,
                              local_map_encoder="resnet", local_map_embedding_dim=400, local_map_size=20,
                              down_dims=[512, 1024, 2048])
    net.load_state_dict(onet.state_dict())
    return onet, net


def make_sampler(net, k=1):
    from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler
    return DiffusionSampler(net, None, "carmaze", policy="flow_matching", pred_horizon=64, action_dim=2,
                            prediction_type="actions", obs_history=1, action_history=1, goal_conditioned=True,
                            num_diffusion_iters=k, local_map_size=20).eval()


def test_car_env_matches_oracle_env():
    from ditreeonlineplanner_amd.car_env import CarEnv
    maze = load_maze("val_maze_7")
    env = CarEnv(maze_map=maze, collision_checking=False)
    ref = ORRT.OracleCarEnv(maze_map=maze, collision_checking=False)
    opts = {"reset_cell": np.array([1, 1]), "reset_deg": 0.0, "goal_cell": np.array([1, 4])}
    env.reset(options=opts)
    ref.reset(options=opts)
    assert np.array_equal(env.goal, ref.goal)
    assert np.array_equal(env.cell_xy_to_rowcol(np.array([0.3, -1.2])), ref.cell_xy_to_rowcol(np.array([0.3, -1.2])))
    rng = np.random.default_rng(0)
    for i in range(80):
        a = np.array([rng.uniform(2, 12), rng.uniform(-0.05, 0.05)])
        o1, r1, t1, _, i1 = env.step(a)
        o2, r2, t2, _, i2 = ref.step(a)
        assert np.abs(o1 - o2).max() < 1e-9 and i1["success"] == i2["success"] and t1 == t2
    assert env.done and ref.done                      # drove straight into the goal cell, then frozen
    assert env.is_done(env.state)
    env.reset_done()
    assert not env.done


def test_lidar_facade_matches_reference_vectors():
    from ditreeonlineplanner_amd.lidar_sim.lidar_2d_sim import Lidar2DSim
    g = golden("geometry")
    maze = load_maze("boxes")
    lid = Lidar2DSim()
    assert lid.scan_time == 0.2 and np.array_equal(lid.angles_deg, G.LIDAR_ANGLES_DEG)
    p = g["lidar_boxes_poses"][3]
    d, e, v = lid.scan(p, maze)
    assert np.abs(d - g["lidar_boxes_dist"][3]).max() < 1e-9 and np.abs(e - g["lidar_boxes_end"][3]).max() < 1e-9
    _, _, v_ref, _ = G.lidar_scan(p, maze)
    assert set(map(tuple, v)) == set(map(tuple, v_ref))
    # the driver's use (run_scenarios_with_lidar_DiTree.py:118-121)
    known = np.zeros_like(maze)
    ends = np.floor(e).astype(int)
    known[ends[:, 1], ends[:, 0]] = 1
    assert (maze[known == 1] == 1).all()


def test_diffusion_sampler_forward(net_pair):
    onet, net = net_pair
    smp = make_sampler(net)
    maze = load_maze("boxes").astype(np.float32)
    rng = np.random.default_rng(4)
    B = 8
    st = np.stack([rng.uniform(-8, 8, B), rng.uniform(-8, 8, B), rng.uniform(-3, 3, B), rng.uniform(0, 4, B),
                   rng.uniform(0, 1, B), rng.uniform(-0.4, 0.4, B)], axis=1)
    prev = np.stack([rng.uniform(-5, 10, B), rng.uniform(-2, 2, B)], axis=1)
    goal = np.array([7.5, 7.5])
    lm = G.create_local_map(maze, st[:, 0], st[:, 1], st[:, 2], 20, 0.2, 1.0, (10.0, 10.0))
    torch.manual_seed(11)
    a = smp(st[:, None, :], prev_actions=prev[:, None, :], goal=goal, local_map=torch.tensor(lm))
    assert a.shape == (B, 64, 2) and a.dtype == np.float64
    torch.manual_seed(11)
    noise = torch.randn((B, 64, 2), device="cuda").cpu().numpy()
    cond = OS.car_cond_vector(st, prev, np.ones(B, bool), goal)
    ref = OS.unnormalize_actions(OS.flow_sample(onet, noise, OS.scale_local_map(lm), cond))
    assert np.linalg.norm(a - ref) / np.linalg.norm(ref) < 2e-2
    # prev_actions=None: zeros stay un-normalised (fm_policy.py:113-122)
    torch.manual_seed(11)
    a0 = smp(st[:, None, :], prev_actions=None, goal=goal, local_map=torch.tensor(lm))
    cond0 = OS.car_cond_vector(st, prev, np.zeros(B, bool), goal)
    ref0 = OS.unnormalize_actions(OS.flow_sample(onet, noise, OS.scale_local_map(lm), cond0))
    assert np.linalg.norm(a0 - ref0) / np.linalg.norm(ref0) < 2e-2


def test_propagate_contract(net_pair):
    """planners/base_planner.py:257-320 return contract: done True/False/None and the slices."""
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    _, net = net_pair
    maze = load_maze("val_maze_7")
    env = CarEnv(maze_map=maze, collision_checking=False)
    start = np.array([*env.cell_rowcol_to_xy(np.array([1, 1])), 0.0, 0.0, 0.0, 0.0])
    goal = np.array([*env.cell_rowcol_to_xy(np.array([1, 4])), 0, 0, 0, 0.0])
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=make_sampler(net), action_horizon=8,
                     local_map_size=20, local_map_scale=0.2, global_map_scale=1.0, prop_duration=[64], time_budget=5,
                     batch=16)
    acts = np.tile(np.array([[8.0, 0.0]]), (8, 1))
    obs, done, a, s = pl.propagate_action_sequence_env(start, acts.copy())
    ref = G.rollout_chunk(start[None], acts[None], maze, env.goal, 8)
    assert done is False and a.shape == (8, 2) and s.shape == (1, 9, 6)
    assert np.abs(obs - ref["end_state"][0]).max() < 1e-9 and np.abs(s[0] - ref["states"][0]).max() < 1e-9
    # collision: turn hard into the wall from a fast state
    fast = start.copy()
    fast[3] = 4.0
    fast[2] = np.pi / 2
    obs, done, a, s = pl.propagate_action_sequence_env(fast, acts.copy())
    ref = G.rollout_chunk(fast[None], acts[None], maze, env.goal, 8)
    assert ref["status"][0] == 2 and done is None
    i = ref["n_steps"][0] - 1
    assert a.shape == (i, 2) and s.shape == (1, i, 6)
    # goal: start just short of the goal radius, drive in
    near = np.array([env.goal[0] - 0.6, env.goal[1], 0.0, 3.0, 0.5, 0.0])
    env.reset_done()
    obs, done, a, s = pl.propagate_action_sequence_env(near, acts.copy())
    ref = G.rollout_chunk(near[None], acts[None], maze, env.goal, 8)
    assert ref["status"][0] == 1 and done is True
    k = ref["n_steps"][0]
    assert (a[k:] == 0).all() and (a[:k] != 0).any() and (s[0, k + 1:] == 0).all()
    assert pl.check_collision(np.array([-3.4, 0.0, 0.0])) == bool(G.is_colliding_car(np.array([[-3.4, 0.0, 0.0]]), maze)[0])


def test_rrt_planner_plan_runs_and_reports(net_pair):
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    import random
    _, net = net_pair
    maze = load_maze("boxes")
    env = CarEnv(maze_map=maze, collision_checking=False)
    start = np.array([*env.cell_rowcol_to_xy(np.array([17, 2])), np.deg2rad(45.0), 0.0, 0.0, 0.0])
    goal = np.array([*env.cell_rowcol_to_xy(np.array([2, 17])), 0, 0, 0, 0.0])
    random.seed(42)
    np.random.seed(42)
    torch.manual_seed(42)
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=make_sampler(net), prediction_type="actions",
                     action_horizon=8, local_map_size=20, local_map_scale=0.2, global_map_scale=1.0,
                     goal_conditioning_bias=0.85, prop_duration=[64], time_budget=120, max_iter=300, verbose=False,
                     batch=64, max_candidates=192)
    pl.reset()
    path, actions = pl.plan()
    r = pl.results
    assert set(["iterations", "time", "path", "actions", "number_of_nodes", "path_time"]) <= set(r)
    n = r["number_of_nodes"]
    assert n >= 1 and r["iterations"] > 0
    if path is not None:
        assert path.dtype == np.float32 and path.shape[1] == 6 and actions.dtype == np.float32
        assert abs(r["path_time"] - len(path) * env.dt) < 1e-12
        # the path ends on a tree node and starts at the start state
        assert np.allclose(path[0], start.astype(np.float32))
    nodes = pl.node_list
    assert len(nodes) == n and nodes[0].parent is None
    for nd in nodes[1:]:
        assert nd.parent is not None and nd.parent_states_seq.shape[0] == 1
        assert np.array_equal(nd.parent_states_seq[0, 0], nd.parent.state)       # edge starts at the parent
        assert np.array_equal(nd.parent_states_seq[0, -1], nd.state)
    # a second plan after reset starts from a fresh tree
    pl.reset(start_state=start, goal_state=goal)
    assert pl._engine.tree.n_nodes_host == 1


def test_planner_with_a_diffusion_policy_sampler(net_pair):
    """policy='diffusion' (policies/fm_policy.py:164-182): with a DDPM scheduler of the reference's configuration
    (run_scenarios.py:157-158) the planner's rounds run the K reverse steps on the device; a scheduler object the device step
    does not implement is refused instead of being sampled with the wrong rule."""
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.ddpm import DDPMScheduler
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler

    class Sch:
        timesteps = []

        def set_timesteps(self, n):
            pass
    _, net = net_pair
    mk = lambda sch, k: DiffusionSampler(net, sch, "carmaze", policy="diffusion", pred_horizon=64, action_dim=2,      # noqa: E731
                                         prediction_type="actions", obs_history=1, action_history=1, goal_conditioned=True,
                                         num_diffusion_iters=k, local_map_size=20)
    maze = load_maze("boxes")
    env = CarEnv(maze_map=maze, collision_checking=False)
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), 0.7, 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    kw = dict(env_id="carmaze", environment=env, action_horizon=8, local_map_size=20, local_map_scale=0.2, global_map_scale=1.0,
              prop_duration=[32])
    with pytest.raises(NotImplementedError, match="DDPM scheduler"):
        RRT_Planner(start, goal, sampler=mk(Sch(), 3), time_budget=1, **kw)
    sch = DDPMScheduler(num_train_timesteps=4, beta_schedule="squaredcos_cap_v2", clip_sample=True, prediction_type="epsilon")
    smp = mk(sch, 4)
    pl = RRT_Planner(start, goal, sampler=smp, time_budget=600, batch=32, max_candidates=64, **kw)
    assert pl._engine.ddpm is not None and pl._engine.ddpm[1].shape == (4, 5)
    random.seed(3)
    np.random.seed(3)
    torch.manual_seed(3)
    path, actions = pl.plan()
    n = pl.results["number_of_nodes"]
    assert pl.results["iterations"] > 0 and n >= 1
    if path is not None:
        assert path.dtype == np.float32 and path.shape[1] == 6 and np.isfinite(path).all()
        # clip_sample: x0 in [-1, 1] -> the un-normalised actions of every edge stay within mean +- a few std
        assert np.abs(actions).max() < 20
    # the sampler's own forward takes the same on-device loop
    a = smp(np.zeros((3, 1, 6)), prev_actions=None, goal=np.array([3.0, 4.0]), local_map=np.zeros((3, 20, 20), dtype=np.float32))
    assert a.shape == (3, 64, 2) and np.isfinite(a).all()


def test_planner_counts_collision_checks_and_defaults_to_f32_class_precision(net_pair):
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.common import map_utils
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    _, net = net_pair
    smp = make_sampler(net)
    assert smp.precision == _lib.PREC_F16X3
    maze = load_maze("boxes")
    env = CarEnv(maze_map=maze, collision_checking=False)
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), 0.7, 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=smp, action_horizon=8, local_map_size=20,
                     local_map_scale=0.2, global_map_scale=1.0, prop_duration=[32, 16], time_budget=60, batch=32,
                     max_candidates=64)
    map_utils.cc_calls = 0
    pl.reset()
    pl.plan()
    steps = int(pl._engine.rb.chunk_steps[:32].sum().item())
    assert map_utils.cc_calls >= steps > 0                  # two rounds of 32 candidates, one test per executed env step
    assert map_utils.is_colliding_car(np.array([-100.0, 0.0, 0.0]), maze) is True          # out of the map
    lm = map_utils.create_local_map(maze, start[0], start[1], start[2], 20, 0.2, 1.0, (10.0, 10.0))
    assert lm.shape == (1, 20, 20) and np.array_equal(lm, G.create_local_map(maze, start[:1], start[1:2], start[2:3], 20, 0.2, 1.0, (10.0, 10.0)))


class TapeSampler:
    """A non-network sampler in the facade's host-sampler protocol (RRT_Planner._host_actions): the golden action tape."""

    def __init__(self, seed):
        from oracle.tapes import ActionTape
        self.tape = ActionTape(seed)

    def sample_round(self, first, B, n_chunks, P):
        return np.stack([self.tape.actions(np.arange(first, first + B), j) for j in range(n_chunks)], axis=1)


def _tape_planner(tag, batch, **kw):
    import random
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    g = golden("traces")
    maze = load_maze(str(g[f"trace_{tag}_maze_name"]))
    sr, sc, sdeg, gr, gc = [int(v) for v in g[f"trace_{tag}_scenario"]]
    start = np.array([*G.cell_rowcol_to_xy([sr, sc], maze), np.deg2rad(float(sdeg)), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([gr, gc], maze), 0, 0, 0, 0])
    env = CarEnv(maze_map=maze, collision_checking=False)
    random.seed(42)
    np.random.seed(42)
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=TapeSampler(int(g[f"trace_{tag}_tape_seed"])),
                     action_horizon=8, local_map_size=20, local_map_scale=0.2, global_map_scale=1.0,
                     goal_conditioning_bias=0.85, prop_duration=[64], time_budget=600, batch=batch,
                     max_candidates=int(g[f"trace_{tag}_budget"]), **kw)
    pl.reset()
    return g, pl


@pytest.mark.parametrize("tag", ["race", "boxes"])
def test_planner_with_a_non_network_sampler_reproduces_the_reference_trace(tag):
    """BASELINE config 1 (plumbing: carmaze + a non-network sampler, Race Track row 1) THROUGH the reference-shaped surface:
    RRT_Planner(batch=1) fed by the golden action tape builds the reference planner's tree (parents bit-exact, states 1e-9)
    and returns its path."""
    g, pl = _tape_planner(tag, 1)
    path, actions = pl.plan()
    snap = pl._engine.tree_snapshot()
    assert np.array_equal(snap["parents"], g[f"trace_{tag}_parents"])
    assert np.abs(snap["states"] - g[f"trace_{tag}_states"]).max() < 1e-9
    assert pl.results["iterations"] == int(g[f"trace_{tag}_iterations"])
    assert (pl._engine.goal_node is not None) == bool(g[f"trace_{tag}_reached"])
    assert path.shape == g[f"trace_{tag}_path"].shape and np.abs(path - g[f"trace_{tag}_path"]).max() < 1e-5
    assert np.array_equal(actions, g[f"trace_{tag}_actions"])
    assert len(pl.node_list) == len(snap["parents"]) and pl.node_list is pl.node_list        # cached between accesses


def test_draw_ahead_leaves_the_global_rngs_where_a_sequential_run_leaves_them():
    """Rounds are pre-drawn on a helper thread from `random` / `np.random`; a pre-drawn round that is never expanded (goal
    reached, budget over) must be un-drawn: after plan() both generators are where batch = 1 (no draw-ahead) leaves them
    after the same number of candidates."""
    import random
    g, pl1 = _tape_planner("boxes", 1)
    pl1.max_candidates = 96
    pl1.plan()
    state1 = (random.getstate(), np.random.get_state()[1].copy(), np.random.get_state()[2])
    g, pl2 = _tape_planner("boxes", 32)
    pl2.max_candidates = None                 # wall-clock budget only: a round IS drawn ahead when the loop ends
    pl2.time_budget = 1e9
    orig = pl2._engine.expand_round
    calls = [0]

    def three_rounds(*a, **k):
        calls[0] += 1
        cnt = orig(*a, **k)
        if calls[0] == 3:
            pl2.time_budget = -1.0            # the budget ends after round 3, with round 4 already drawn
        return cnt
    pl2._engine.expand_round = three_rounds
    pl2.plan()
    assert calls[0] == 3
    state2 = (random.getstate(), np.random.get_state()[1].copy(), np.random.get_state()[2])
    assert state1[0] == state2[0] and np.array_equal(state1[1], state2[1]) and state1[2] == state2[2]


def test_uniform_sampler_through_the_planner():
    """policies/uniform_policy.UniformSampler(env.action_space) as the planner's sampler (the wiring the reference left
    commented out, run_scenarios.py:275-283): rounds run on injected uniform actions, the tree does not depend on the
    round size, and the one-action-per-call form of the reference class works as well."""
    import random
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    from ditreeonlineplanner_amd.policies.uniform_policy import UniformSampler
    maze = load_maze("boxes")
    env = CarEnv(maze_map=maze, collision_checking=False)
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    trees = []
    for batch in (1, 16):
        random.seed(42)
        np.random.seed(42)
        pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=UniformSampler(env.action_space, seed=3),
                         action_horizon=8, local_map_size=20, local_map_scale=0.2, global_map_scale=1.0, prop_duration=[64],
                         time_budget=600, batch=batch, max_candidates=16, early_exit=bool(batch > 1))
        pl.reset()
        pl.plan()
        trees.append(pl._engine.tree_snapshot())
        assert pl.results["iterations"] > 0
    # 16 rounds of 1 vs one round of 16: same samples and actions; a round expands against the tree of its start, so the
    # trees agree up to the first candidate whose nearest node was appended inside the round -- the root's children do
    assert trees[0]["states"][1].tolist() == trees[1]["states"][1].tolist()

    class OneAction:                                          # the reference class's call shape: (1, action_dim)
        def __call__(self, *a, **k):
            return np.array([[4.0, 0.1]])
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=OneAction(), action_horizon=8, local_map_size=20,
                     local_map_scale=0.2, global_map_scale=1.0, prop_duration=[16], time_budget=600, batch=4, max_candidates=4)
    pl.reset()
    pl.plan()
    a = pl._engine.rb.actions[:4].cpu().numpy()
    ran = pl._engine.rb.chunk_steps[:4].cpu().numpy()
    assert (a[ran > 0][:, 0] == np.array([4.0, 0.1])).all()
