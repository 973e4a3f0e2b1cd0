"""RRT_Planner(env_id='antmaze') -- the reference-shaped surface of BASELINE config 3 (run_scenarios.py:225-233,241-251) with a
CALLER-SUPPLIED env step: the planner steps the caller's env object (ant_env.set_state + step, planners/base_planner.py:278-298)
between the two halves of every chunk, everything else runs on the device."""
import random

import numpy as np
import pytest
import torch

from oracle import ant as OA
from oracle import rrt as ORRT
from tests.test_oracle_ant import trace_setup

pytestmark = pytest.mark.gpu


class MazeData:
    def __init__(self, maze, s):
        self.maze_map, self.maze_size_scaling = np.asarray(maze), s
        self.map_length, self.map_width = self.maze_map.shape
        self.x_map_center, self.y_map_center = self.map_width / 2 * s, self.map_length / 2 * s

    def cell_xy_to_rowcol(self, xy):
        return np.array([np.floor((self.y_map_center - xy[1]) / self.maze_size_scaling),
                         np.floor((xy[0] + self.x_map_center) / self.maze_size_scaling)])


class AntEnv:
    """A caller's env: the gym surface the planner touches, stepping the numpy restatement of the stand-in model (a real
    deployment puts MuJoCo here)."""

    def __init__(self, maze, s_global, desired):
        self.maze_data = MazeData(maze, s_global)
        self.ant_env = self
        self.desired = np.asarray(desired, dtype=np.float64)
        self.state = np.zeros(29)
        self.n_steps = self.n_set = 0

    def _obs(self):
        return {"achieved_goal": self.state[:2].copy(), "desired_goal": self.desired.copy(), "observation": self.state[2:].copy()}

    def reset(self, options=None, **kw):
        return self._obs(), {}

    def set_state(self, qpos, qvel):
        self.state = np.concatenate([qpos, qvel]).astype(np.float64)
        self.n_set += 1

    def step(self, action):
        self.state = OA.ant_model_step(self.state[None], np.asarray(action, dtype=np.float64)[None])[0]
        self.n_steps += 1
        return self._obs(), 0.0, False, False, {}


class TapeSampler:
    """A non-network sampler: action sequences as a pure function of the global candidate index (the golden trace's tape)."""

    def __init__(self, tape):
        self.tape = tape

    def sample_round(self, first, B, n_chunks, P):
        cand = np.arange(first, first + B)
        return np.stack([self.tape.actions(cand, j) for j in range(n_chunks)], axis=1)


def _planner(g, pre, m, env, sampler, **kw):
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    return RRT_Planner(g[pre + "start"], g[pre + "goal"], env_id="antmaze", environment=env, sampler=sampler,
                       prediction_type="actions", action_horizon=2, local_map_size=16, local_map_scale=0.8, global_map_scale=4.0,
                       goal_conditioning_bias=0.85, prop_duration=[48], time_budget=1e9, max_iter=300, verbose=False, **kw)


@pytest.mark.parametrize("tag", ["model_boxes", "model_val7"])
def test_antmaze_planner_with_a_caller_supplied_step_builds_the_reference_tree(tag):
    """batch = 1, the golden action tape as the sampler, the env stepped on the host: the facade builds the tree of the
    reference's RRT_Planner(env_id='antmaze') (golden trace; the reference ran on the same numpy env): parents exact, node
    states exact, iterations, path and actions equal."""
    g, pre, pl, atape, otape, m = trace_setup(tag)
    env = AntEnv(m["maze"], 4.0, g[pre + "desired"])
    planner = _planner(g, pre, m, env, TapeSampler(atape), batch=1, max_candidates=m["candidates"], capacity=1024)
    random.seed(42)
    np.random.seed(42)
    path, actions = planner.plan()
    nodes = planner.node_list
    idx = {id(n): i for i, n in enumerate(nodes)}
    parents = np.array([-1 if n.parent is None else idx[id(n.parent)] for n in nodes])
    assert np.array_equal(parents, g[pre + "parents"])
    assert np.array_equal(np.array([n.state for n in nodes]), g[pre + "states"])        # host-stepped: the same numpy arithmetic
    assert planner.results["iterations"] == m["iterations"] and planner.results["number_of_nodes"] == len(parents)
    assert np.array_equal(path, g[pre + "path"]) and np.array_equal(actions, g[pre + "actions"])
    assert env.n_steps > 0 and env.n_set > 0
    # a child's parent_states_seq / parent_action_seq are the filtered edge rows (RRT.py:196-200)
    n5 = nodes[min(5, len(nodes) - 1)]
    assert n5.parent_states_seq.shape[0] == 1 and n5.parent_states_seq.shape[2] == 29 and n5.parent_action_seq.shape[1] == 8
    # the reference-named single-edge call, through the same env + the device collision test
    st, done, acts, seq = planner.propagate_action_sequence_env(g[pre + "start"].copy(), atape.actions(np.array([0]), 0)[0, :2])
    assert seq.shape[0] == 1 and seq.shape[2] == 29 and (done is None or done in (True, False))
    assert planner.check_collision(g[pre + "start"]) is False
    up = g[pre + "start"].copy()
    up[3:7] = [0.0, 1.0, 0.0, 0.0]                                                     # upside down (map_utils.py:126-133)
    assert planner.check_collision(up) is True


def test_antmaze_planner_with_the_network_and_on_device_dynamics():
    """DiffusionSampler(env_id='antmaze') + ant_dynamics='model' (the stand-in on the device) and 'host' (the same model
    stepped by the caller): same seeds -> same tree (1e-9); rounds of 64."""
    from ditreeonlineplanner_amd._lib import PREC_F32
    from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler
    from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net
    g, pre, pl, atape, otape, m = trace_setup("model_boxes")
    torch.manual_seed(0)
    net = init_noise_pred_net(input_dim=8, action_dim=8, obs_dim=29, obs_history=3, action_history=1, goal_conditioned=True,
                              goal_dim=2, local_map_conditioned=True, local_map_encoder="resnet", local_map_embedding_dim=400,
                              local_map_size=16, down_dims=[512, 1024, 2048])
    smp = DiffusionSampler(net, None, "antmaze", policy="flow_matching", pred_horizon=16, action_dim=8, prediction_type="actions",
                           obs_history=3, action_history=1, goal_conditioned=True, num_diffusion_iters=1, local_map_size=16,
                           precision=PREC_F32)
    trees = []
    for dyn in ("model", "host"):
        env = AntEnv(m["maze"], 4.0, g[pre + "desired"])
        planner = _planner(g, pre, m, env, smp, batch=64, max_candidates=128, capacity=1024, ant_dynamics=dyn)
        random.seed(7)
        np.random.seed(7)
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        path, actions = planner.plan()
        snap = planner._engine.tree_snapshot()
        trees.append((snap["parents"], snap["states"], path, actions))
        assert path is not None and path.shape[1] == 29 and actions.shape[1] == 8 and path.dtype == np.float32
        assert planner.results["number_of_nodes"] == len(snap["parents"]) > 3
        assert (env.n_steps > 0) == (dyn == "host")
    assert np.array_equal(trees[0][0], trees[1][0])
    assert np.abs(trees[0][1] - trees[1][1]).max() < 1e-9 and np.abs(trees[0][2] - trees[1][2]).max() < 1e-5
