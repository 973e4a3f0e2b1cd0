"""`MPPI.mppi.MPPI` -- the controller `run_scenarios_with_lidar_MPPI.py:10,339-449` constructs as
`MPPI(maze_data, T, K, nx, nu)` and drives through `.reset / .step / .is_done / .set_ref_path / .reference_path /
.update_maze / .env` (`:392-394,402,410,414,417,422,442,447-449`).

The reference repository does NOT contain this module ("no oracle exists at all"), so there are no
semantics to be faithful to: the controller below is the build's own -- information-theoretic MPPI (Williams et al. 2017)
on the reference's car dynamics, two-ball collision test and goal radius, every controller step on the GPU
(`ditree_mppi_step`: K x T rollouts with on-device noise, soft-min weights by wavefront reductions, weighted control update,
one executed env step).  The cost function is stated in include/ditree.h and DESIGN.md; **parity with the reference is
unpinned**, the kernels are held to the build's numpy restatement (oracle/mppi.py) and to invariants
(tests/test_gpu_mppi.py).  There is no CPU path.

`step(state) -> (next_state, action, done)` keeps the driver's contract: `done is None` when the executed step collides
with the KNOWN maze (the state does not advance, the nominal controls restart from rest and the next call draws fresh
noise: the driver retries up to ALLOWED_TRIALS times), `True` inside the goal radius, else `False`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .car_env import CarEnv
from .ops import default_context


class MPPI:
    def __new__(cls, maze_data=None, T=16, K=1024, nx=6, nu=2, *a, **kw):
        # MPPI(maze_data, T, K, nx=29, nu=8): BASELINE config 5's antmaze variant -- the same controller on the higher-DoF slot
        if cls is MPPI and (nx, nu) == (29, 8):
            return super().__new__(AntMPPI)
        return super().__new__(cls)

    def __init__(self, maze_data=None, T=16, K=1024, nx=6, nu=2, lam=1.0, sigma=(3.0, 0.6), w_track=20.0, w_progress=0.5,
                 w_collision=1.0e3, w_goal=50.0, window_back=8, window_fwd=56, seed=0, lanes=0, ctx=None, env=None,
                 rank=None, world_size=None, process_group=None, **kw):
        if maze_data is None:
            raise ValueError("MPPI needs the known maze (maze_data)")
        if nx != 6 or nu != 2:
            raise NotImplementedError("the controller drives the car model: nx = 6 (x, y, psi, v, D, delta), nu = 2 (dD, ddelta)")
        if not (1 <= int(T) <= 64) or int(K) < 1:
            raise ValueError("1 <= T <= 64, K >= 1")
        self.T, self.K, self.nx, self.nu = int(T), int(K), nx, nu
        # K is the GLOBAL number of rollouts; with world_size > 1 (one process per GPU) every rank runs a contiguous block
        from .engine import default_shard
        d_rank, d_world, d_pg = default_shard()
        self.world = int(d_world if world_size is None else world_size)
        self.rank = int((d_rank if self.world == d_world else 0) if rank is None else rank)
        self.pg = d_pg if process_group is None else process_group
        per = (self.K + self.world - 1) // self.world
        self.k_lo = min(self.rank * per, self.K)
        self.K_local = max(min(self.k_lo + per, self.K) - self.k_lo, 0)
        if self.K_local < 1:
            raise ValueError("fewer rollouts than ranks")
        self.ctx = ctx or default_context()
        self.env = env if env is not None else CarEnv(maze_map=np.asarray(maze_data), collision_checking=False, ctx=self.ctx)
        self.maze = np.float32(maze_data)
        self.params = _lib.MppiParams(self.T, self.K_local, float(lam), (C.c_double * 2)(float(sigma[0]), float(sigma[1])),
                                      float(w_track), float(w_progress), float(w_collision), float(w_goal), int(seed),
                                      int(window_back), int(window_fwd), int(lanes), int(self.k_lo))
        dev = self.ctx.device
        f64 = torch.float64
        self._state = torch.zeros(6, dtype=f64, device=dev)
        self._U = torch.zeros(self.T, 2, dtype=f64, device=dev)
        self._costs = torch.zeros(self.K_local, dtype=f64, device=dev)
        self._flags = torch.zeros(self.K_local, dtype=torch.int32, device=dev)
        self._result = torch.zeros(16, dtype=f64, device=dev)
        self._state_host = None            # the state the device holds (skip the upload when the driver hands it back)
        self._sums = torch.zeros(3 + 2 * self.T, dtype=f64, device=dev)
        self._path = None
        self.reference_path = None
        self.goal_state = None
        self.counter = 0                   # one noise stream per step() call (also the retried ones)
        self.last = {}
        self.ctx.upload_maze(self.maze, owner=self)

    # ------------------------------------------------------------------ the surface the driver uses
    def reset(self, start_state=None, goal_state=None):
        if start_state is not None:
            start_state = np.asarray(start_state, dtype=np.float64)
            self.goal_state = np.asarray(goal_state, dtype=np.float64)
            opts = {"reset_cell": self.env.cell_xy_to_rowcol(start_state[:2]), "reset_deg": np.rad2deg(start_state[2]),
                    "goal_cell": self.env.cell_xy_to_rowcol(self.goal_state[:2])}
            self.env.reset(options=opts)
            self.env.set_state(start_state.copy())
        self._U.zero_()
        self.counter = 0

    def update_maze(self, new_maze):
        self.maze = np.float32(new_maze)
        self.env.maze_map = new_maze
        self.ctx.upload_maze(self.maze, owner=self)

    def set_ref_path(self, path):
        """The plan to track: (P, >= 2) states; more than 4096 points are thinned uniformly (the kernel stages the path in LDS)."""
        path = np.asarray(path, dtype=np.float64)
        if path.ndim != 2 or path.shape[1] < 2 or len(path) < 1:
            raise ValueError("reference path must be (P, >= 2)")
        self.reference_path = path
        xy = np.ascontiguousarray(path[:, :2])
        if len(xy) > 4096:
            xy = np.ascontiguousarray(xy[np.round(np.linspace(0, len(xy) - 1, 4096)).astype(int)])
        self._path = torch.as_tensor(xy, device=self.ctx.device)

    def is_done(self, state):
        return self.env.is_done(state)

    def step(self, state, noise=None):
        """One controller step from `state` (6,): -> (next_state (6,) f64, action (2,) f64, done in {False, True, None})."""
        if self._path is None:
            raise _lib.DitreeError("MPPI.step: set_ref_path(path) first")
        if self.ctx.maze_owner is not self:
            self.ctx.upload_maze(self.maze, owner=self)
        state = np.asarray(state, dtype=np.float64)
        if self._state_host is None or not np.array_equal(state, self._state_host):
            self._state.copy_(torch.as_tensor(state))
        self.controller_step(noise=noise)
        res = self._result.cpu().numpy()            # the one D2H (and sync) of a step: action, status, statistics, new state
        self.counter += 1
        status = int(res[2])
        self.last = {"beta": float(res[3]), "eta": float(res[4]), "nearest_path_index": int(res[5]),
                     "collided_rollouts": int(res[6]), "effective_samples": float(res[7])}
        nxt = res[8:14].copy()
        self._state_host = nxt.copy()
        action = res[:2].copy()
        if status == 2:
            return nxt, action, None
        self.env.set_state(nxt.copy())
        if status == 1:
            self.env.done = True
        return nxt, action, status == 1

    def controller_step(self, noise=None, weights=None):
        """One controller step on the device state / controls.  One rank: a single C-ABI call.  Sharded: rollouts + local
        minimum, all-reduce(MIN) of beta, weighted sums, all-reduce(SUM) of 3 + 2T doubles, then the (replicated) update and
        executed step -- every rank ends with the same state and controls."""
        if self.world <= 1:
            return self.launch(_lib.MPPI_ALL, noise=noise, weights=weights)
        import torch.distributed as dist
        gloo = dist.get_backend(self.pg) == "gloo"

        def allreduce(t, op):
            if gloo:
                h = t.cpu()
                dist.all_reduce(h, op=op, group=self.pg)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op, group=self.pg)
        self.launch(_lib.MPPI_ROLLOUTS | _lib.MPPI_MIN, noise=noise)
        allreduce(self._result[3:4], dist.ReduceOp.MIN)
        self.launch(_lib.MPPI_SUMS, noise=noise, weights=weights)
        allreduce(self._sums, dist.ReduceOp.SUM)
        self.launch(_lib.MPPI_APPLY | _lib.MPPI_EXECUTE, noise=noise, weights=weights)

    # ------------------------------------------------------------------ one C-ABI call (stages: _lib.MPPI_*)
    def launch(self, stages, noise=None, weights=None):
        goal = (C.c_double * 2)(float(self.env.goal[0]), float(self.env.goal[1]))
        _lib.check(self.ctx._h, _lib.lib().ditree_mppi_step(
            self.ctx._h, C.byref(self.params), self._state.data_ptr(), self._U.data_ptr(), self._path.data_ptr(),
            int(self._path.shape[0]), goal, None if noise is None else noise.data_ptr(), self.counter, int(stages),
            self._costs.data_ptr(), None if weights is None else weights.data_ptr(), self._flags.data_ptr(),
            self._sums.data_ptr(), self._result.data_ptr(), self.ctx.stream), "mppi_step")


class _AntGoalEnv:
    """What the driver reads from ``mppi.env`` for the ant: the goal and the state (no physics here: the rollouts and the
    executed step run the build's stand-in model on the device)."""

    def __init__(self, maze, s_global):
        self.maze_map, self.s_global = np.asarray(maze), float(s_global)
        self.goal = np.zeros(2)
        self.state = np.zeros(29)
        self.done = False

    def set_state(self, state):
        self.state = np.asarray(state, dtype=np.float64)

    def is_done(self, state):
        d = np.asarray(state, dtype=np.float64)[:2] - self.goal
        return bool(np.linalg.norm(d) < 0.45 * self.s_global)            # planners/base_planner.py:296-297


class AntMPPI(MPPI):
    """`MPPI(maze_data, T, K, nx=29, nu=8)`: BASELINE config 5 as written ("MPPI antmaze, 65 536 rollouts") -- the build's MPPI
    controller (above) on the higher-DoF rollout slot: every rollout step is one env step of the build's STAND-IN crawler model
    (include/ditree.h ditree_ant_model; NOT MuJoCo), followed by the reference's ant collision test (common/map_utils.py:
    126-219) and goal radius (planners/base_planner.py:296-297).  The reference has neither an MPPI module nor the ant's
    physics in its repository: PARITY UNPINNED BY CONSTRUCTION; held to oracle/mppi.py (tests/test_gpu_mppi.py)."""

    def __init__(self, maze_data=None, T=16, K=1024, nx=29, nu=8, lam=1.0, sigma=0.5, w_track=2.0, w_progress=0.5, w_collision=1.0e3,
                 w_goal=50.0, window_back=8, window_fwd=56, seed=0, ctx=None, s_global=4.0, ball_radius=1.2, model=None,
                 rank=None, world_size=None, process_group=None, **kw):
        if maze_data is None:
            raise ValueError("MPPI needs the known maze (maze_data)")
        if (nx, nu) != (29, 8):
            raise NotImplementedError("AntMPPI: nx = 29, nu = 8")
        if not (1 <= int(T) <= 64) or int(K) < 1:
            raise ValueError("1 <= T <= 64, K >= 1")
        self.T, self.K, self.nx, self.nu = int(T), int(K), 29, 8
        from .engine import default_shard
        d_rank, d_world, d_pg = default_shard()
        self.world = int(d_world if world_size is None else world_size)
        self.rank = int((d_rank if self.world == d_world else 0) if rank is None else rank)
        self.pg = d_pg if process_group is None else process_group
        per = (self.K + self.world - 1) // self.world
        self.k_lo = min(self.rank * per, self.K)
        self.K_local = max(min(self.k_lo + per, self.K) - self.k_lo, 0)
        if self.K_local < 1:
            raise ValueError("fewer rollouts than ranks")
        self.ctx = ctx or default_context()
        self.s_global = float(s_global)
        self.env = _AntGoalEnv(maze_data, s_global)
        self.maze = np.float32(maze_data)
        sg = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (8,))
        self.params = _lib.MppiAntParams(self.T, self.K_local, float(lam), (C.c_double * 8)(*[float(v) for v in sg]), float(w_track),
                                         float(w_progress), float(w_collision), float(w_goal), int(seed), int(window_back),
                                         int(window_fwd), int(self.k_lo), 0.45 * float(s_global), float(ball_radius), float(s_global),
                                         _lib.AntModel.default() if model is None else model)
        dev = self.ctx.device
        f64 = torch.float64
        self._state = torch.zeros(29, dtype=f64, device=dev)
        self._U = torch.zeros(self.T, 8, dtype=f64, device=dev)
        self._costs = torch.zeros(self.K_local, dtype=f64, device=dev)
        self._flags = torch.zeros(self.K_local, dtype=torch.int32, device=dev)
        self._result = torch.zeros(64, dtype=f64, device=dev)
        self._state_host = None
        self._sums = torch.zeros(3 + 8 * self.T, dtype=f64, device=dev)
        self._path = None
        self.reference_path = None
        self.goal_state = None
        self.counter = 0
        self.last = {}
        self.ctx.upload_maze(self.maze, owner=self)

    def reset(self, start_state=None, goal_state=None, desired_goal=None):
        if start_state is not None:
            self.goal_state = np.asarray(goal_state, dtype=np.float64)
            self.env.goal = np.asarray(self.goal_state[:2] if desired_goal is None else desired_goal, dtype=np.float64)[:2].copy()
            self.env.set_state(np.asarray(start_state, dtype=np.float64).copy())
            self.env.done = False
        self._U.zero_()
        self.counter = 0

    def update_maze(self, new_maze):
        self.maze = np.float32(new_maze)
        self.env.maze_map = np.asarray(new_maze)
        self.ctx.upload_maze(self.maze, owner=self)

    def step(self, state, noise=None):
        """One controller step from `state` (29,): -> (next_state (29,), action (8,), done in {False, True, None})."""
        if self._path is None:
            raise _lib.DitreeError("MPPI.step: set_ref_path(path) first")
        if self.ctx.maze_owner is not self:
            self.ctx.upload_maze(self.maze, owner=self)
        state = np.asarray(state, dtype=np.float64)
        if self._state_host is None or not np.array_equal(state, self._state_host):
            self._state.copy_(torch.as_tensor(state))
        self.controller_step(noise=noise)
        res = self._result.cpu().numpy()
        self.counter += 1
        status = int(res[2])
        self.last = {"beta": float(res[3]), "eta": float(res[4]), "nearest_path_index": int(res[5]),
                     "collided_rollouts": int(res[6]), "effective_samples": float(res[7])}
        nxt = res[24:53].copy()
        self._state_host = nxt.copy()
        action = res[16:24].copy()
        if status == 2:
            return nxt, action, None
        self.env.set_state(nxt.copy())
        if status == 1:
            self.env.done = True
        return nxt, action, status == 1

    def launch(self, stages, noise=None, weights=None):
        goal = (C.c_double * 2)(float(self.env.goal[0]), float(self.env.goal[1]))
        _lib.check(self.ctx._h, _lib.lib().ditree_mppi_step_ant(
            self.ctx._h, C.byref(self.params), self._state.data_ptr(), self._U.data_ptr(), self._path.data_ptr(),
            int(self._path.shape[0]), goal, None if noise is None else noise.data_ptr(), self.counter, int(stages),
            self._costs.data_ptr(), None if weights is None else weights.data_ptr(), self._flags.data_ptr(),
            self._sums.data_ptr(), self._result.data_ptr(), self.ctx.stream), "mppi_step_ant")
