"""BasePlanner / Node facade (reference: planners/base_planner.py:24-363) over the HIP engine.

The tree lives on the GPU (engine.DeviceTree); ``Node`` objects are only materialised on demand
for callers that walk ``node_list``.  ``propagate_action_sequence_env`` keeps the reference's
return contract -- ``(obs, done, actions, states[None])`` with ``done is None`` on collision --
and runs the fused dynamics + goal + collision kernel.
"""
from __future__ import annotations

import abc
import random
import time
from collections import deque

import numpy as np
import torch

from ..engine import ExpansionEngine
from ..ops import default_context


class Node:
    def __init__(self, state, parent_action_seq=None, parent_states_seq=None, parent=None):
        self.state = state
        self.parent_action_seq = parent_action_seq
        self.parent_states_seq = parent_states_seq
        self.parent = parent
        self.cached_actions = deque([])
        self.num_visit = 0

    def __repr__(self):
        return f"Node(State={self.state}, nVisits={self.num_visit})"


class BasePlanner(abc.ABC):
    def __init__(self, start_state, goal_state, environment, sampler, action_horizon=8, local_map_size=(10, 10),
                 local_map_scale=0.2, global_map_scale=1.0, env_id="pushT", time_budget=10, **kwargs):
        if environment is None:
            raise ValueError("Environment is not defined.")
        self.is_ant = "ant" in env_id.lower()
        if "car" not in env_id.lower() and not self.is_ant:
            raise NotImplementedError("the engine covers the car and the ant environments (pushT / pointmaze / dronemaze: not built)")
        self.env = environment
        self.device = "cuda"
        self.sampler = sampler
        self.action_horizon = action_horizon
        self.local_map_size = local_map_size
        self.local_map_scale = local_map_scale
        self.s_global = global_map_scale
        self.env_id = env_id
        self.time_budget = time_budget
        self.start_node = Node(np.asarray(start_state, dtype=np.float64))
        self.goal_state = np.asarray(goal_state, dtype=np.float64)
        self.results = {"iterations": 0, "time": 0, "path": None, "actions": None, "number_of_nodes": 0}
        self.render = kwargs.get("render", False)
        self.verbose = kwargs.get("verbose", False)
        self.env_dt = self.env.dt if hasattr(self.env, "dt") else 0.1
        self.max_v = 5
        if self.is_ant:
            # planners/base_planner.py:81-92: the maze geometry comes from env.maze_data (gymnasium-robotics Maze: centres
            # already scaled by maze_size_scaling = global_map_scale)
            md = self.env.maze_data
            start = md.cell_xy_to_rowcol(start_state[:2])
            goal = md.cell_xy_to_rowcol(goal_state[:2])
            self.options = {"reset_cell": start, "goal_cell": goal}
            self._reset_out = self.env.reset(options=self.options)
            self.x_center, self.y_center = md.x_map_center, md.y_map_center
            self.map_width, self.map_length = md.map_width, md.map_length
            self.maze = np.float32(md.maze_map)
        else:
            start = self.env.cell_xy_to_rowcol(start_state[:2])
            goal = self.env.cell_xy_to_rowcol(goal_state[:2])
            self.options = {"reset_cell": start, "reset_deg": np.rad2deg(start_state[2]), "goal_cell": goal}
            self.env.reset(options=self.options)
            self.x_center = self.env.x_map_center
            self.y_center = self.env.y_map_center
            self.map_width = len(self.env.maze_map[0])
            self.map_length = len(self.env.maze_map)
            self.maze = np.float32(self.env.maze_map)
        self.save_bad_edges = False
        self.failed_node_list = []
        self._debug = kwargs.get("debug", False)
        self._scenario_num = str(kwargs.get("scenario_num", "999"))
        self._scenario_name = str(kwargs.get("scenario_name", "test"))
        self.scenario_iter_num = str(kwargs.get("iter_num", "0"))
        self.save_path = str(kwargs.get("root_folder", "benchmark_results"))
        self.ctx = kwargs.get("ctx") or getattr(sampler, "ctx", None) or default_context()
        self._engine: ExpansionEngine | None = None

    @property
    def scenario_iter_folder_name(self):
        return f"Iter_{self.scenario_iter_num}"

    @abc.abstractmethod
    def plan(self):
        pass

    @abc.abstractmethod
    def reset(self):
        pass

    # ------------------------------------------------------------------ collision / propagation
    def check_collision(self, state=None):
        """is_colliding_car(state, maze) (common/map_utils.py:103-115) on the device: a zero-velocity
        one-step rollout leaves the pose unchanged, so its collision flag is the pose's."""
        ctx = self.ctx
        ctx.upload_maze(self.maze)
        if self.is_ant:
            # is_colliding_ant(state, self.maze, 1.2, self.s_global) (planners/base_planner.py:154-155)
            st = torch.as_tensor(np.asarray(state, dtype=np.float64).reshape(1, -1).copy(), device=ctx.device)
            return bool(ctx.ant_collision(st, 1.2, self.s_global)[0].item())
        st = np.zeros((1, 6))
        st[0, :3] = np.asarray(state, dtype=np.float64)[:3]
        s = torch.as_tensor(st, device=ctx.device)
        a = torch.zeros(1, 1, 2, dtype=torch.float64, device=ctx.device)
        status, _, _, _ = ctx.car_rollout(s, a, np.array([1e9, 1e9]), A=1)
        return (int(status.item()) & 0xFF) == 2

    def propagate_action_sequence_env(self, state, action_sequence):
        """planners/base_planner.py:257-320."""
        if action_sequence is None:
            raise ValueError("Action sequence is None.")
        if self.is_ant:
            return self._propagate_ant(np.asarray(state, dtype=np.float64), np.array(action_sequence, dtype=np.float64))
        ctx = self.ctx
        A = self.action_horizon
        action_sequence = np.asarray(action_sequence, dtype=np.float64)
        n = min(len(action_sequence), A)
        state = np.asarray(state, dtype=np.float64)
        self.env.set_state(state)
        if getattr(self.env, "done", False) or getattr(self.env, "terminated", False):
            # latched env (car_env.py:254): the step is frozen and reports success
            states = np.zeros((A + 1, 6))
            states[0] = state
            states[1] = state
            if self.check_collision(state):
                return state, None, action_sequence[:0], states[:0][None, :]
            acts = action_sequence[:A].copy()
            acts[1:] = 0
            return state, True, acts, states[: len(acts) + 1][None, :]
        ctx.upload_maze(self.maze)
        s = torch.as_tensor(state.reshape(1, 6).copy(), device=ctx.device)
        a = torch.as_tensor(np.ascontiguousarray(action_sequence[:n]).reshape(1, n, 2), device=ctx.device)
        status, states, aout, steps = ctx.car_rollout(s, a, np.asarray(self.env.goal, dtype=np.float64), A=n)
        code = int(status.item())
        k = int(steps.item())
        obs = s.cpu().numpy()[0]
        st_full = np.zeros((A + 1, 6))
        st_full[: n + 1] = states.cpu().numpy()[0]
        self.env.set_state(obs)
        if (code & 0xFF) == 2:
            if code & 0x100:
                self.env.done = True
            i = k - 1
            return obs, None, action_sequence[:i], st_full[:i][None, :]
        done = (code & 0xFF) == 1
        if done:
            self.env.done = True
        acts = aout.cpu().numpy()[0]
        return obs, done, acts, st_full[: len(acts) + 1][None, :]

    def ant_obs(self, step_result):
        """planners/base_planner.py:298: the 29-d planner state of an env step's observation dict."""
        obs = step_result[0] if isinstance(step_result, tuple) else step_result
        return np.hstack((obs["achieved_goal"], obs["observation"])).astype(np.float64), np.asarray(obs["desired_goal"], dtype=np.float64)

    def _propagate_ant(self, state, action_sequence):
        """planners/base_planner.py:257-320, ant branches: the env step is the CALLER's simulator (MuJoCo in the reference;
        nothing of it is built here), the goal test (:296-297) runs on the host value and check_collision (:154-155,306) on the
        device (ditree_ant_collision)."""
        A = self.action_horizon
        self.env.ant_env.set_state(state[:15], state[15:])
        states = np.zeros((A + 1, state.shape[0]))
        states[0] = state
        obs, done = state, False
        for i in range(len(action_sequence[:A])):
            obs, desired = self.ant_obs(self.env.step(action_sequence[i]))
            d = obs[:2] - desired
            done = bool(np.linalg.norm(d) < 0.45 * self.s_global)
            states[i + 1] = obs
            if self.check_collision(obs):
                return obs, None, action_sequence[:i], states[:i][None, :]
            if done:
                action_sequence[i + 1:] = 0
                break
        action_sequence = action_sequence[:A]
        return obs, done, action_sequence, states[: len(action_sequence) + 1][None, :]

    # ------------------------------------------------------------------ sampling (host, reference RNG order)
    def sample_row_col_from_probability_map(self):
        """planners/base_planner.py:157-160: one categorical draw over the env's sampling-probability map."""
        pm = self.env.prob_map
        flat = np.random.choice(pm.size, size=1, p=pm.ravel())
        row, col = np.unravel_index(flat, pm.shape)
        return row[np.newaxis], col[np.newaxis]

    def random_node_sample(self, batch_size=1):
        """planners/base_planner.py:162-207 (car): python ``random`` then ``np.random``; run_type >= 2 draws the
        position as the centre of a cell of the sampling-probability map instead of uniformly."""
        if random.random() > self.goal_sample_rate:
            if "ant" in self.env_id.lower():                      # planners/base_planner.py:193-200
                state = np.zeros((1, 29))
                x = np.random.uniform(-self.s_global * self.map_width / 2, self.s_global * self.map_width / 2, size=(batch_size, 1))
                y = np.random.uniform(-self.s_global * self.map_length / 2, self.s_global * self.map_length / 2, size=(batch_size, 1))
                state[:, :2] = np.concatenate((x, y), axis=1)
                return state
            if getattr(self, "run_type", 0) >= 2:
                rows, cols = self.sample_row_col_from_probability_map()
                x, y = self.env.cell_rowcol_to_xy(np.array([rows[0], cols[0]]))
                x, y = x[np.newaxis], y[np.newaxis]
            else:
                x = np.random.uniform(-self.map_width / 2, self.map_width / 2, size=(batch_size, 1))
                y = np.random.uniform(-self.map_length / 2, self.map_length / 2, size=(batch_size, 1))
            theta = np.random.uniform(-np.pi, np.pi, size=(batch_size, 1))
            v = np.random.uniform(-self.max_v, self.max_v, size=(batch_size, 1))
            throttle = np.random.uniform(-1, 1, size=(batch_size, 1))
            steer = np.random.uniform(-0.40, 0.40, size=(batch_size, 1))
            return np.concatenate((x, y, theta, v, throttle, steer), axis=1)
        sample = np.zeros((batch_size, self.start_node.state.shape[0]))
        sample[:] = self.goal_state
        return sample

    def handle_goal_reached(self, node_index, iterations, start_time):
        self.results["time"] = time.time() - start_time
        path, actions = self._engine.path_to(node_index)
        self.results["iterations"] = iterations
        self.results["path"] = path
        self.results["path_time"] = len(path) * self.env_dt
        self.results["actions"] = actions
        self.results["number_of_nodes"] = self._engine.tree.n_nodes_host
        return path, actions

    def handle_goal_not_reached(self, iterations, start_time):
        self.results["time"] = time.time() - start_time
        self.results["iterations"] = iterations
        self.results["number_of_nodes"] = self._engine.tree.n_nodes_host
        return None, None

    def visualize_tree(self, *a, **k):           # plotting is reporting, outside the hot-path scope
        pass
