"""Round-based restatement of the reference's RRT expand loop (car, run_type 0).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference sites: planners/RRT.py:113-257 (plan), :49-51 (nearest),
planners/base_planner.py:162-207 (random_node_sample), :231-244 (goal bookkeeping),
:257-320 (propagate), :342-363 (path walk), car_env.py:206-238 (reset -> env.goal is
the *centre of the goal cell*).

Semantics of one round of B candidates (B = 1 is the reference loop, step for step):
  1. draw B samples from the tape in candidate order (RandomTape mirrors the
     reference's RNG call order: random.random(), 6 x np.random.uniform when not a
     goal sample, then random.random() for the conditioning coin);
  2. nearest node of every sample against the tree *as of the round start*;
  3. every candidate rolls out edge_length / action_horizon chunks
     (local map -> sampler -> 8 env steps with goal / collision tests);
  4. candidates are accepted in candidate-index order: a collided edge is dropped,
     the others become nodes N, N+1, ...; the lowest accepted index that reached the
     goal ends the plan and later candidates of that round are dropped.

"Sticky done": when an env step lands inside the goal radius *and* collides, the
reference leaves ``env.done`` latched (car_env.py:254,266; nothing resets it before the
next candidate).  The next candidate's first env step is then frozen and reports
success, so the reference appends a zero-length edge and returns "goal reached".
``emulate_sticky_done=True`` reproduces that defect; traces used for parity assert it
never triggered.
"""
from __future__ import annotations

import random as _pyrandom

import numpy as np

from . import geometry as G


class RandomTape:
    """The reference's two global RNG streams (``random`` and ``np.random``) as objects."""

    def __init__(self, seed: int = 42):
        self.py = _pyrandom.Random(seed)
        self.np = np.random.RandomState(seed)

    def _xy(self, width, length, prob_map):
        """base_planner.py:181-187: uniform over the map, or (run_type >= 2) the centre of a cell drawn from
        the sampling-probability map."""
        if prob_map is None:
            x = self.np.uniform(-width / 2, width / 2, size=(1, 1))
            y = self.np.uniform(-length / 2, length / 2, size=(1, 1))
            return x, y
        from .prob_maps import draw_cell
        row, col = draw_cell(self.np, prob_map)
        xy = G.cell_rowcol_to_xy(np.array([row, col]), prob_map)
        return np.array([[xy[0]]]), np.array([[xy[1]]])

    def draw_candidate(self, width, length, goal_state, goal_sample_rate=0.15,
                       goal_conditioning_bias=0.85, max_v=5.0, prob_map=None):
        """base_planner.py:162-207 + RRT.py:153-156 -> (sample (6,), cond_goal_xy (2,))."""
        if self.py.random() > goal_sample_rate:
            x, y = self._xy(width, length, prob_map)
            th = self.np.uniform(-np.pi, np.pi, size=(1, 1))
            v = self.np.uniform(-max_v, max_v, size=(1, 1))
            thr = self.np.uniform(-1, 1, size=(1, 1))
            st = self.np.uniform(-0.40, 0.40, size=(1, 1))
            sample = np.concatenate((x, y, th, v, thr, st), axis=1)[0]
        else:
            sample = np.array(goal_state, dtype=np.float64).copy()
        if self.py.random() > goal_conditioning_bias:
            cond = sample[:2].copy()
        else:
            cond = np.asarray(goal_state, dtype=np.float64)[:2].copy()
        return sample, cond

    def draw_candidate_ref(self, remain_path, width, length, goal_state, goal_sample_rate=0.15, max_v=5.0,
                           prob_map=None):
        """run_type > 0 (RRT.py:134-140,153-156): with a remaining reference path, pick one of its points
        (np.random.choice) and keep it with probability 0.6, else a random_node_sample(); the
        conditioning goal is always the sample's xy."""
        if remain_path is not None:
            node_idx = self.np.choice(np.arange(len(remain_path)))
            explore = self.py.random() < 0.4
        else:
            node_idx, explore = None, True
        if explore:
            if self.py.random() > goal_sample_rate:
                x, y = self._xy(width, length, prob_map)
                th = self.np.uniform(-np.pi, np.pi, size=(1, 1))
                v = self.np.uniform(-max_v, max_v, size=(1, 1))
                thr = self.np.uniform(-1, 1, size=(1, 1))
                st = self.np.uniform(-0.40, 0.40, size=(1, 1))
                sample = np.concatenate((x, y, th, v, thr, st), axis=1)[0]
            else:
                sample = np.array(goal_state, dtype=np.float64).copy()
        else:
            sample = np.zeros(6)
            sample[:2] = remain_path[node_idx]            # float32 path point (only xy is ever used)
        return sample, sample[:2].copy()

    def draw_round(self, B, width, length, goal_state, **kw):
        s = np.zeros((B, 6))
        c = np.zeros((B, 2))
        for i in range(B):
            s[i], c[i] = self.draw_candidate(width, length, goal_state, **kw)
        return s, c


def cell_center_f32(xy, maze):
    """env.reset(options) puts the env state at the centre of the start cell in a float32 array
    (car_env.py:215-229); planner.reset derives that cell from start_state (RRT.py:38-40)."""
    rc = G.cell_xy_to_rowcol(np.asarray(xy, dtype=np.float64), maze)
    return G.cell_rowcol_to_xy(rc, maze).astype(np.float32)


def path_after_obstacle(init_main_path, cur_xy, maze):
    """planners/RRT.py:83-111 extract_path_after_obstacle: the xy of ``init_main_path`` (float32, as generate_final_path_env
    returns it) from the point nearest to ``cur_xy`` = env.state[:2] onwards, cut to what lies behind the first blocked
    stretch.  numpy's dtype rules are part of the function: ``curr_state - path`` is float64 when the env state is float64
    (the f32 path is promoted; the norm and its argmin are then float64) and float32 when the state is float32, as right after
    env.reset; the cell lookup (cell_xy_to_rowcol of every f32 path point) stays float32 either way."""
    maze = np.asarray(maze)
    P = np.asarray(init_main_path)[:, :2]
    cur = np.asarray(cur_xy)
    cur = cur[:2].astype(np.float32) if cur.dtype == np.float32 else cur[:2].astype(np.float64)
    closest = int(np.argmin(np.linalg.norm(cur - P, axis=1)))                       # RRT.py:86-87
    rest = P[closest:].copy()
    H, W = maze.shape
    if rest.dtype == np.float32:
        rc = np.stack([np.floor((np.float32(H / 2) - rest[:, 1]) / np.float32(1)), np.floor((rest[:, 0] + np.float32(W / 2)) / np.float32(1))],
                      axis=1).astype(int)
    else:
        rc = np.stack([np.floor((H / 2 - rest[:, 1]) / 1), np.floor((rest[:, 0] + W / 2) / 1)], axis=1).astype(int)
    k = -1
    for i, pnt in enumerate(rc):
        if maze[pnt[0], pnt[1]] == 1:
            k = i
            break
    cp = rc[k]
    while maze[cp[0], cp[1]] == 1 and k < len(rc):
        cp = rc[k]
        k += 1
    return rest[k:]


class OracleTree:
    def __init__(self, start_state, n_chunks, A):
        self.states = [np.asarray(start_state, dtype=np.float64).copy()]
        self.parents = [-1]
        self.last_action = [np.zeros(2)]
        self.has_prev = [False]
        self.num_visit = [0]
        # per node: edge states (k, 6) and actions (m, 2) after the zero-row filter (RRT.py:196-199)
        self.edge_states = [None]
        self.edge_actions = [None]

    def __len__(self):
        return len(self.states)

    def xy(self):
        return np.asarray(self.states)[:, :2]


class OraclePlanner:
    """Batched-round RRT.  ``sampler(cand_idx, chunk_idx, state (n,6), prev_action (n,2),
    has_prev (n,), cond_goal (n,2), local_map (n,N,N)) -> actions (n, >=A, 2) float64``
    where cand_idx are *global* candidate indices (the action tape's key)."""

    def __init__(self, maze, start_state, goal_state, sampler, edge_length=64, action_horizon=8,
                 local_map_size=20, local_map_scale=0.2, s_global=1.0, goal_sample_rate=0.15,
                 goal_conditioning_bias=0.85, emulate_sticky_done=True, run_type=0, init_main_path=None,
                 prop_duration=None):
        self.maze = np.asarray(maze, dtype=np.float32)
        self.start_state = np.asarray(start_state, dtype=np.float64)
        self.goal_state = np.asarray(goal_state, dtype=np.float64)
        # planner.reset -> env.reset(options): goal = centre of the goal cell (car_env.py:225-226)
        self.env_goal = G.cell_rowcol_to_xy(G.cell_xy_to_rowcol(self.goal_state[:2], self.maze), self.maze)
        self.sampler = sampler
        # planners/RRT.py:26,149-152: the edge length of a visit is prop_duration[clip(parent.num_visit)], and every visit
        # increments parent.num_visit.  In a round the candidates of one parent are visits in candidate order: candidate b
        # sees num_visit + (number of earlier candidates of the round with the same parent).
        self.schedule = [int(edge_length)] if prop_duration is None else [int(v) for v in prop_duration]
        edge_length = max(self.schedule)
        self.H = edge_length
        self.A = action_horizon
        self.n_chunks = edge_length // action_horizon
        self.lm_size, self.lm_scale, self.s_global = local_map_size, local_map_scale, s_global
        self.center = G.map_center(self.maze, 1.0)          # base_planner.py:113-114 (maze_size_scaling 1)
        self.gsr, self.gcb = goal_sample_rate, goal_conditioning_bias
        self.tree = OracleTree(self.start_state, self.n_chunks, self.A)
        self.sticky = emulate_sticky_done
        self.env_done_latched = False
        self.sticky_triggered = False
        self.run_type = run_type
        self.init_main_path = None if init_main_path is None else np.asarray(init_main_path)
        self.obstacle_ahead = []       # per node 1.. (RRT.py:202-205)
        self.iterations = 0            # the reference's iter_num (chunk iterations)
        self.candidates = 0
        self.goal_node = None

    # ------------------------------------------------------------------ one round
    def expand_round(self, samples, cond_goals):
        samples = np.asarray(samples, dtype=np.float64)
        cond_goals = np.asarray(cond_goals, dtype=np.float64)
        B = samples.shape[0]
        t = self.tree
        parent = G.nn_argmin(samples[:, :2], t.xy())
        cur = np.asarray(t.states)[parent].copy()
        prev_a = np.asarray(t.last_action)[parent].copy()
        has_prev = np.asarray(t.has_prev)[parent].copy()
        cand_idx = np.arange(self.candidates, self.candidates + B)
        A, nC = self.A, self.n_chunks
        all_states = np.zeros((B, nC, A + 1, 6))
        all_actions = np.zeros((B, nC, A, 2))
        chunk_status = np.full((B, nC), -1, dtype=np.int32)       # -1 = chunk not run
        chunk_steps = np.zeros((B, nC), dtype=np.int32)
        alive = np.ones(B, dtype=bool)
        budget = np.full(B, nC, dtype=np.int32)
        if len(self.schedule) > 1:
            seen = {}
            for b in range(B):
                k = t.num_visit[parent[b]] + seen.get(int(parent[b]), 0)
                seen[int(parent[b])] = seen.get(int(parent[b]), 0) + 1
                budget[b] = self.schedule[min(max(k, 0), len(self.schedule) - 1)] // A
        final_status = np.zeros(B, dtype=np.int32)
        g_at_c = np.zeros(B, dtype=bool)
        chunks_run = np.zeros(B, dtype=np.int32)
        for j in range(nC):
            idx = np.nonzero(alive)[0]
            if idx.size == 0:
                break
            lm = G.create_local_map(self.maze, cur[idx, 0], cur[idx, 1], cur[idx, 2], self.lm_size,
                                    self.lm_scale, self.s_global, self.center)
            acts = np.asarray(self.sampler(cand_idx[idx], j, cur[idx], prev_a[idx], has_prev[idx],
                                           cond_goals[idx], lm), dtype=np.float64)[:, :A]
            r = G.rollout_chunk(cur[idx], acts, self.maze, self.env_goal, A)
            all_states[idx, j] = r["states"]
            all_actions[idx, j] = r["actions"]
            chunk_status[idx, j] = r["status"]
            chunk_steps[idx, j] = r["n_steps"]
            chunks_run[idx] += 1
            cur[idx] = r["end_state"]
            g_at_c[idx] |= r["goal_at_collision"]
            final_status[idx] = r["status"]
            ok = r["status"] == G.STATUS_OK
            # prev_actions = curr_action_seq -> last row (RRT.py:188); only continuing candidates matter
            prev_a[idx[ok]] = r["actions"][ok, A - 1]
            has_prev[idx[ok]] = True
            alive[idx] = ok & (j + 1 < budget[idx])
        # ---------------------------------------------------------- accept in candidate order
        accepted = []
        for b in range(B):
            self.candidates += 1
            t.num_visit[parent[b]] += 1
            if self.sticky and self.env_done_latched:
                # frozen env: one chunk, one frozen step reporting success (see module docstring)
                self.sticky_triggered = True
                self.iterations += 1
                s0 = np.asarray(t.states)[parent[b]]
                node = self._append(s0.copy(), parent[b], np.stack([s0, s0]), all_actions[b, 0, :1].copy())
                accepted.append(node)
                self.obstacle_ahead.append(bool(G.check_obstacle_ahead(s0, self.maze)[0]) if self.run_type > 0 else False)
                self.goal_node = node
                break
            self.iterations += int(chunks_run[b])
            if final_status[b] == G.STATUS_COLLIDED:
                if g_at_c[b]:
                    self.env_done_latched = True
                continue
            es, ea = [], []
            for j in range(chunks_run[b]):
                es.append(all_states[b, j])
                ea.append(all_actions[b, j])
            es = np.concatenate(es)
            ea = np.concatenate(ea)
            ea = ea[~(ea == 0).all(axis=1)]                          # RRT.py:196-197
            es = es[~(es == 0).all(axis=1)]                          # RRT.py:198-199
            node = self._append(cur[b].copy(), parent[b], es, ea)
            accepted.append(node)
            self.obstacle_ahead.append(bool(G.check_obstacle_ahead(cur[b], self.maze)[0]) if self.run_type > 0 else False)
            if final_status[b] == G.STATUS_GOAL:
                self.goal_node = node
                break
        return dict(parent=parent, status=final_status, end_state=cur, accepted=accepted,
                    chunks_run=chunks_run, chunk_status=chunk_status, chunk_steps=chunk_steps,
                    states=all_states, actions=all_actions, goal_at_collision=g_at_c)

    def _append(self, state, parent, edge_states, edge_actions):
        t = self.tree
        t.states.append(state)
        t.parents.append(int(parent))
        t.edge_states.append(edge_states)
        t.edge_actions.append(edge_actions)
        if len(edge_actions) > 0:
            t.last_action.append(edge_actions[-1].copy())
            t.has_prev.append(True)
        else:
            t.last_action.append(np.zeros(2))
            t.has_prev.append(False)
        t.num_visit.append(0)
        return len(t.states) - 1

    # ------------------------------------------------------------------ driver
    def remaining_reference_path(self):
        """RRT.py:83-111 extract_path_after_obstacle seen from the env state right after planner.reset (= centre of the
        start cell, a FLOAT32 array: car_env.py:215-229)."""
        return path_after_obstacle(self.init_main_path, cell_center_f32(self.start_state[:2], self.maze), self.maze)

    def plan(self, tape: RandomTape, n_candidates: int, batch: int = 1):
        """Run rounds of ``batch`` candidates until the goal is reached or ``n_candidates``
        were drawn (the reference's wall-clock budget replaced by a candidate budget)."""
        H, W = self.maze.shape
        remain = None
        if self.run_type > 0 and self.init_main_path is not None:
            remain = self.remaining_reference_path()
        pm = self.sampling_map()
        while self.goal_node is None and self.candidates < n_candidates:
            B = min(batch, n_candidates - self.candidates)
            if self.run_type == 0:
                s, c = tape.draw_round(B, W, H, self.goal_state, goal_sample_rate=self.gsr,
                                       goal_conditioning_bias=self.gcb, prob_map=pm)
            else:
                s, c = np.zeros((B, 6)), np.zeros((B, 2))
                for i in range(B):
                    s[i], c[i] = tape.draw_candidate_ref(remain, W, H, self.goal_state, goal_sample_rate=self.gsr,
                                                         prob_map=pm)
            self.expand_round(s, c)
        reached = self.goal_node is not None
        if reached:
            node = self.goal_node
        elif self.run_type > 0 and all(self.obstacle_ahead):          # np.all([]) is True (RRT.py:227-232)
            return False, None, None
        else:
            node = self.fallback_node()
        path, actions = self.path_to(node) if node is not None else (None, None)
        return reached, path, actions

    def sampling_map(self):
        """car_env.py:100-137 + RRT.py:124-125: None for run_type < 2, the EDT prior for run_type 2, the prior
        log-blended with the start -> goal Gaussian (cells as (col, row)) for run_type >= 3."""
        if self.run_type < 2:
            return None
        from . import prob_maps as PM
        prior = PM.edt_prior(np.asarray(self.maze, dtype=np.float64))
        if self.run_type == 2:
            return prior
        start_rc = G.cell_xy_to_rowcol(cell_center_f32(self.start_state[:2], self.maze), self.maze)
        goal_rc = G.cell_xy_to_rowcol(self.env_goal, self.maze)
        gauss, _, _ = PM.gaussian_map(start_rc[::-1], goal_rc[::-1])
        return PM.combine_log_blend(prior, gauss)

    def fallback_node(self):
        """RRT.py:233-254: nearest to the goal (+1e4 when an obstacle is ahead), or -- with a reference
        path and run_type > 0 -- the node that got furthest along init_main_path."""
        if len(self.tree) < 2:
            return None
        xy = self.tree.xy()[1:]
        obs = np.array(self.obstacle_ahead, dtype=bool) if self.run_type > 0 else np.zeros(len(xy), dtype=bool)
        if self.run_type == 0 or self.init_main_path is None:
            d = xy - self.goal_state[:2]
            cost = G.norm2(d[:, 0], d[:, 1]) + 10e3 * obs.astype(int)
            return 1 + int(np.argmin(cost))
        P = np.asarray(self.init_main_path)[:, :2]
        prog = np.array([int(np.argmin(np.linalg.norm(q - P, axis=1))) if not o else -1 for q, o in zip(xy, obs)])
        return 1 + int(np.argmax(prog))

    def path_to(self, node):
        """base_planner.py:342-363: float32 concatenation of edge states + node states."""
        t = self.tree
        path, actions = [], []
        n = node
        while n != -1:
            seg = [t.states[n]]
            if t.edge_states[n] is not None:
                seg = list(t.edge_states[n]) + seg
            path = seg + path
            if t.edge_actions[n] is not None:
                actions = list(t.edge_actions[n]) + actions
            n = t.parents[n]
        path = np.array(path, dtype=np.float32) if path else None
        actions = np.array(actions, dtype=np.float32) if actions else None
        return path, actions


# ---------------------------------------------------------------------- CarEnv restatement
class OracleCarEnv:
    """The subset of car_env.py:21-396 the planner and drivers touch (run_type 0).

    Used (a) by tests as the CPU reference for the product's ``CarEnv`` facade and (b) by
    tests/golden/make_golden.py as the environment the *reference* planner is run against
    (car_env.py itself needs casadi + gymnasium, absent here).
    """

    def __init__(self, maze_map, collision_checking=False, run_type=0):
        self._maze_map = np.asarray(maze_map)
        self.dt = G.CAR_DT
        self.collision_checking = collision_checking
        self.run_type = run_type
        self.goal = np.array([0, 0])
        self.done = False
        self.terminated = False
        self._state = np.zeros(6)
        self.current_step = 0
        # car_env.py:100-110 (the constructor passes (row, col) where the later updates pass (col, row))
        from . import prob_maps as PM
        self.prior = PM.edt_prior(self._maze_map)
        if run_type < 2:
            self.prob_map = np.zeros_like(self._maze_map.copy())
        elif run_type == 2:
            self.prob_map = self.prior
        else:
            g, _, _ = PM.gaussian_map(self.cell_xy_to_rowcol(self.state[:2]), self.cell_xy_to_rowcol(self.goal[:2]))
            self.prob_map = PM.combine_log_blend(self.prior, g)

    @property
    def maze_map(self):
        return self._maze_map

    @maze_map.setter
    def maze_map(self, m):
        """car_env.py:117-128."""
        from . import prob_maps as PM
        self._maze_map = m
        self.prior = PM.edt_prior(m)
        if self.run_type == 2:
            self.prob_map = self.prior
        elif self.run_type >= 3:
            self.update_prob_map_by_loc()

    def update_prob_map_by_loc(self):
        """car_env.py:130-137."""
        from . import prob_maps as PM
        g, _, _ = PM.gaussian_map(self.cell_xy_to_rowcol(self.state[:2])[::-1], self.cell_xy_to_rowcol(self.goal[:2])[::-1])
        self.prob_map = PM.combine_log_blend(self.prior, g)

    @property
    def x_map_center(self):
        return self._maze_map.shape[1] / 2

    @property
    def y_map_center(self):
        return self._maze_map.shape[0] / 2

    @property
    def state(self):
        return self._state.copy()

    def cell_rowcol_to_xy(self, rc):
        rc = np.asarray(rc)                       # car_env.py:189-194 indexes the FIRST axis ((2, 1) inputs occur)
        xc, yc = G.map_center(self._maze_map)
        return np.array([(rc[1] + 0.5) * 1.0 - xc, yc - (rc[0] + 0.5) * 1.0])

    def cell_xy_to_rowcol(self, xy, floor_enable=True):
        return G.cell_xy_to_rowcol(xy, self._maze_map, floor_enable=floor_enable)

    def reset(self, *, seed=None, options=None, **kw):
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

    def is_done(self, state):
        return bool(G.goal_reached(np.asarray(state, dtype=np.float64), self.goal))

    def reset_done(self):
        self.done = False

    def step(self, action):
        collision = False
        if not self.done and not self.terminated:
            self._state = G.car_step(np.asarray(self._state, dtype=np.float64), np.asarray(action))
            self.current_step += 1
            self.done = bool(G.goal_reached(self._state, self.goal))
            if self.collision_checking:
                collision = bool(G.is_colliding_car(self._state[None, :3], self._maze_map)[0])
            reward = 0
            if collision:
                reward = -1.0
                self.terminated = True
        else:
            reward = 0.0
        info = {"collision": collision, "goal": self.goal, "success": self.done}
        return self.state, reward, self.terminated, False, info
