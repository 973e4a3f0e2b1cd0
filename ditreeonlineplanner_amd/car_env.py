"""CarEnv facade -- the surface of the reference's car_env.py:21-396 that the planner and the
run_scenarios*.py drivers touch, with the per-step arithmetic on the GPU.

State (x, y, psi, v, D, delta), action (dD, ddelta), explicit Euler at 50 Hz, success when
||xy - goal|| < 0.5 where goal is the *centre of the goal cell*; ``done`` / ``terminated`` latch
and freeze the state exactly like car_env.py:254,274-275.  ``step`` runs the HIP rollout kernel
(B = 1, A = 1); there is no CPU dynamics path.  gymnasium / casadi are not required.
"""
from __future__ import annotations

import numpy as np
import torch

from .lidar_sim.lidar_2d_sim import Lidar2DSim
from .ops import default_context
from .prob_sampling_utils import combine_log_blend, edt_prior, gaussian_map


class _Box:
    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(np.float32)


class CarEnv:
    def __init__(self, lidar2dsim: Lidar2DSim | None = None, dt=0.02, drone_radius=0.1, maze_map=None,
                 collision_checking=True, run_type=0, ctx=None):
        if maze_map is None:
            raise ValueError("maze_map is required")
        self.lidar2dsim = lidar2dsim if lidar2dsim is not None else Lidar2DSim()
        self.dt = 1.0 / 50.0
        self.current_step = 0
        self.collision_checking = collision_checking
        self.ball_radius = drone_radius
        self.state_dim, self.action_dim = 6, 2
        self.action_space = _Box([-10.0, -2.0], [10.0, 2.0])          # car_env.py:56-66,590-597
        self._maze_map = np.asarray(maze_map)
        self._maze_size_scaling = 1
        self.goal = np.array([0, 0])
        self.done = False
        self.terminated = False
        self.run_type = run_type
        self._state = np.zeros(6)
        self._ctx = ctx
        self._maze_version = 0
        # sampling-probability map per run_type (car_env.py:100-110; the constructor passes (row, col) to
        # gaussian_map where the later updates pass (col, row))
        self.prior = edt_prior(self._maze_map)
        if run_type < 2:
            self.prob_map = np.zeros_like(self._maze_map.copy())
        elif run_type == 2:
            self.prob_map = self.prior
        else:
            self.gaussian_pdf, _, _ = gaussian_map(self.cell_xy_to_rowcol(self.state[:2]),
                                                   self.cell_xy_to_rowcol(self.goal[:2]))
            self.prob_map = combine_log_blend(self.prior, self.gaussian_pdf)

    # ------------------------------------------------------------------ maze / geometry helpers
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    @property
    def maze_map(self):
        return self._maze_map

    @maze_map.setter
    def maze_map(self, new_maze_map):
        """car_env.py:117-128: a new known maze also refreshes the EDT prior and the sampling map."""
        self._maze_map = np.asarray(new_maze_map)
        self._maze_version += 1
        self.prior = edt_prior(self._maze_map)
        if self.run_type == 2:
            self.prob_map = self.prior
        elif self.run_type >= 3:
            self.update_prob_map_by_loc()

    @property
    def maze_size_scaling(self):
        return self._maze_size_scaling

    @property
    def x_map_center(self):
        return self._maze_map.shape[1] / 2 * self._maze_size_scaling

    @property
    def y_map_center(self):
        return self._maze_map.shape[0] / 2 * self._maze_size_scaling

    @property
    def state(self):
        return np.array(self._state, copy=True)

    def cell_rowcol_to_xy(self, rowcol_pos):
        x = (rowcol_pos[1] + 0.5) * self.maze_size_scaling - self.x_map_center
        y = self.y_map_center - (rowcol_pos[0] + 0.5) * self.maze_size_scaling
        return np.array([x, y])

    def cell_xy_to_rowcol(self, xy_pos, floor_enable=True):
        i = (self.y_map_center - xy_pos[1]) / self.maze_size_scaling
        j = (xy_pos[0] + self.x_map_center) / self.maze_size_scaling
        ret = np.array([i, j])
        return np.floor(ret) if floor_enable else ret

    def update_prob_map_by_loc(self):
        """car_env.py:130-137: prior log-blended with the Gaussian from the current cell to the goal cell."""
        here = self.cell_xy_to_rowcol(self.state[:2])[::-1]
        there = self.cell_xy_to_rowcol(self.goal[:2])[::-1]
        self.gaussian_pdf, _, _ = gaussian_map(here, there)
        self.prob_map = combine_log_blend(self.prior, self.gaussian_pdf)

    # ------------------------------------------------------------------ gym-like API
    def reset(self, *, seed=None, options=None, **kwargs):
        self._state = np.zeros(6, dtype=np.float32)
        if options is not None:
            if options.get("goal_cell") is not None:
                self.goal = self.cell_rowcol_to_xy(options["goal_cell"])
            if options.get("reset_cell") is not None:
                self._state[0:2] = self.cell_rowcol_to_xy(options["reset_cell"])
            if options.get("reset_deg") is not None:
                self._state[2] = np.deg2rad(options["reset_deg"])
        self.current_step = 0
        self.done = False
        self.terminated = False
        return self.state, None

    def set_state(self, state):
        self._state = state

    def reset_done(self):
        self.done = False

    def is_done(self, curr_state):
        d = np.asarray(curr_state, dtype=np.float64)[:2] - np.asarray(self.goal, dtype=np.float64)
        return bool(np.linalg.norm(d) < 0.5)

    def _kernel_step(self, state, action):
        """One env step on the device: returns (next_state, success, collided)."""
        ctx = self.ctx
        ctx.upload_maze(np.asarray(self._maze_map, dtype=np.float32))
        st = torch.as_tensor(np.asarray(state, dtype=np.float64).reshape(1, 6), device=ctx.device)
        act = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 1, 2), device=ctx.device)
        status, _, _, _ = ctx.car_rollout(st, act, np.asarray(self.goal, dtype=np.float64), A=1)
        code = int(status.item())
        nxt = st.cpu().numpy()[0]
        collided = (code & 0xFF) == 2
        success = (code & 0xFF) == 1 or bool(code & 0x100)
        return nxt, success, collided

    def step(self, action):
        collision = False
        if not self.done and not self.terminated:
            nxt, success, collided = self._kernel_step(self._state, action)
            self._state = nxt
            reward = 0
            self.current_step += 1
            self.done = success
            if self.collision_checking:
                collision = collided
            if collision:
                reward = -1.0
                self.terminated = True
        else:
            reward = 0.0
        info = {"collision": collision, "goal": self.goal, "success": self.done}
        return self.state, reward, self.terminated, False, info
