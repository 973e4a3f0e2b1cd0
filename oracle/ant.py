"""Ant (antmaze) glue of the reference's expand loop, restated on the CPU -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

What the reference holds IN ITS OWN SOURCE for the ant (BASELINE config 3) and what this file restates:

  * ``is_colliding_ant``  common/map_utils.py:126-136 (upside-down test on the body z axis, common/se3_utils.py:155-164)
  * ``is_colliding_maze`` common/map_utils.py:139-219 (ONE ball; early returns; the corner rule ignores out-of-map cells,
                          unlike the car's is_colliding_parallel), called as ``(state, maze, 1.2, s_global)``
                          (planners/base_planner.py:154-155)
  * goal test             planners/base_planner.py:296-298: ``||achieved - desired|| < 0.45 * s_global``
  * sampling              planners/base_planner.py:193-200 (zeros(1, 29) with x, y uniform over the scaled map)
  * the chunk loop        planners/RRT.py:131-219 with 29-d states, 8-d actions, obs_history 3
                          (``prev_states = curr_states_seq``; a child's first call sees its parent's filtered edge)

Pinned by goldens written by tests/golden/make_golden.py from the reference's own functions (``antcol_*``: poses x mazes x
quaternions) and from the reference's ``RRT_Planner(env_id='antmaze')`` run on a stand-in env (``anttrace_*``).

What it does NOT restate: the ant's DYNAMICS.  They are MuJoCo 3.1.6 behind gymnasium-robotics 1.3.1 ``AntMaze_Large-v4``
(requirements.txt:59,117; call sites planners/base_planner.py:278-279,290) -- third party, not vendored, not importable
here.  Two stand-ins take their place and both say so wherever they are used:

  * a next-observation TAPE (a pure function of (candidate, chunk, step)), and
  * ``ant_model_step``: THIS BUILD'S OWN surrogate of a four-legged crawler with the ant's interface (29-d observation,
    8-d action in [-1, 1], frame_skip 5 x 0.01 s).  **Not MuJoCo, parity unpinned by construction**; it exists so the
    higher-DoF rollout kernel has arithmetic of the right shape to run, and is held to this numpy restatement at 1e-9.
"""
from __future__ import annotations

import numpy as np

from . import geometry as G

S_DIM, A_DIM = 29, 8
ANT_BALL_RADIUS = 1.2            # planners/base_planner.py:155 "default antmaze values"
ANT_GOAL_FACTOR = 0.45           # planners/base_planner.py:297


# --------------------------------------------------------------------------- collision (reference-pinned)
def body_z_up(q):
    """R[2][2] of common/se3_utils.py:155-164 with ``qw, qx, qy, qz = q``: 1 - 2 (qx^2 + qy^2)."""
    q = np.asarray(q, dtype=np.float64)
    qx, qy = q[..., 1], q[..., 2]
    return 1 - 2 * (qx * qx + qy * qy)


def is_colliding_maze(x, y, maze, s=1.0, r=0.1):
    """common/map_utils.py:139-219 for arrays x, y (one ball each) -> bool array.

    The function returns at its first hit, and every test is side-effect free, so the OR of all tests is its value.  Cells
    are compared with ``== 1`` (:177,184,191,198,214).  Out of the map -> True (:172-173); a NaN coordinate floors to
    INT64_MIN and lands there too.  Corner cells outside the map are skipped (:212-213)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    maze = np.asarray(maze)
    H, W = maze.shape
    xc = W / 2 * s                                                   # :155-156
    yc = H / 2 * s
    with np.errstate(invalid="ignore"):
        row = np.floor((yc - y) / s).astype(np.int64)                # :158-159
        col = np.floor((x + xc) / s).astype(np.int64)
    cell_x = (col + 0.5) * s - xc                                    # :161-162
    cell_y = yc - (row + 0.5) * s
    x_min, x_max = cell_x - s / 2, cell_x + s / 2                    # :165-168
    y_min, y_max = cell_y - s / 2, cell_y + s / 2
    oob = ~((0 <= row) & (row < H)) | ~((0 <= col) & (col < W))       # :171
    rs, cs = np.clip(row, 0, H - 1), np.clip(col, 0, W - 1)

    def occ(ri, ci):
        ok = (ri >= 0) & (ri < H) & (ci >= 0) & (ci < W)
        return ok, maze[np.clip(ri, 0, H - 1), np.clip(ci, 0, W - 1)] == 1

    coll = oob.copy()
    ok, o = occ(rs, cs + 1)
    coll |= ((x + r) > x_max) & (~ok | o)                             # :175-178 right: beyond the map counts as a wall
    ok, o = occ(rs, cs - 1)
    coll |= ((x - r) < x_min) & (~ok | o)                             # :181-185 left
    ok, o = occ(rs - 1, cs)
    coll |= ((y + r) > y_max) & (~ok | o)                             # :188-192 top
    ok, o = occ(rs + 1, cs)
    coll |= ((y - r) < y_min) & (~ok | o)                             # :195-199 bottom
    for cx, cy, ri, ci in ((x_max, y_max, rs - 1, cs + 1), (x_min, y_max, rs - 1, cs - 1),
                           (x_max, y_min, rs + 1, cs + 1), (x_min, y_min, rs + 1, cs - 1)):   # :202-215
        dx, dy = cx - x, cy - y
        dist = np.sqrt(dx * dx + dy * dy)                             # math.sqrt(a ** 2 + b ** 2)
        ok, o = occ(ri, ci)
        coll |= (dist < r) & ok & o
    return coll | oob


def is_colliding_ant(state, maze, ant_radius=ANT_BALL_RADIUS, map_scale=1.0):
    """common/map_utils.py:126-136 for (..., >= 7) states [x, y, z, qw, qx, qy, qz, ...]."""
    state = np.asarray(state, dtype=np.float64)
    up = body_z_up(state[..., 3:7])
    return (up < 0) | is_colliding_maze(state[..., 0], state[..., 1], maze, map_scale, ant_radius)


def ant_goal_reached(obs, desired_xy, s_global):
    """planners/base_planner.py:296-297: np.linalg.norm(achieved_goal - desired_goal) < 0.45 * s_global."""
    d = np.asarray(obs, dtype=np.float64)[..., :2] - np.asarray(desired_xy, dtype=np.float64)
    return G.norm2(d[..., 0], d[..., 1]) < ANT_GOAL_FACTOR * s_global


# --------------------------------------------------------------------------- stand-in dynamics (the BUILD's own; not MuJoCo)
class AntModel:
    """Constants of the surrogate.  include/ditree.h ``ditree_ant_model`` carries the same numbers to the device."""
    h = 0.01                 # integration step [s]
    frame_skip = 5           # sub-steps per env step (gymnasium Ant: frame_skip 5, timestep 0.01)
    k_act, k_spr, k_dmp = 60.0, 20.0, 6.0           # actuator gain, joint spring to rest, joint damping
    k_lim = 400.0            # soft joint stops (continuous: no velocity reset)
    hip_lim = 0.5236         # |hip| <= 30 deg
    ank_lo, ank_hi, ank_rest = 0.5236, 1.2217, 0.87
    contact_gain = 6.0       # foot load = 0.5 (1 + tanh(gain (ankle - rest)))
    leg_r = 0.4              # foot lever arm [m]
    k_push, c_lin = 16.0, 3.0
    z0, z_gain, k_z, c_z = 0.55, 0.3, 120.0, 12.0
    k_lift, c_ang, k_up, k_yaw = 4.0, 5.0, 25.0, 10.0
    # leg mount angles 45, 135, 225, 315 deg
    cphi = (np.sqrt(0.5), -np.sqrt(0.5), -np.sqrt(0.5), np.sqrt(0.5))
    sphi = (np.sqrt(0.5), np.sqrt(0.5), -np.sqrt(0.5), -np.sqrt(0.5))

    @classmethod
    def as_vector(cls):
        """The 24 doubles of ditree_ant_model, in header order."""
        return np.array([cls.h, cls.frame_skip, cls.k_act, cls.k_spr, cls.k_dmp, cls.k_lim, cls.hip_lim, cls.ank_lo, cls.ank_hi,
                         cls.ank_rest, cls.contact_gain, cls.leg_r, cls.k_push, cls.c_lin, cls.z0, cls.z_gain, cls.k_z, cls.c_z,
                         cls.k_lift, cls.c_ang, cls.k_up, cls.k_yaw, cls.cphi[0], cls.sphi[0]], dtype=np.float64)


def ant_model_step(state, action, m=AntModel):
    """One env step (frame_skip sub-steps) of the surrogate for (B, 29) states and (B, 8) actions -> (B, 29).

    Layout as the planner's ant state (planners/base_planner.py:278-279,298): [x, y | z, qw, qx, qy, qz, 8 joints |
    vx, vy, vz, wx, wy, wz, 8 joint velocities]; joints = (hip, ankle) x 4 legs.  Every expression is written in the order the
    device function ``ant_model_substep`` (csrc/ant_device.h) evaluates it; nothing is fused."""
    s = np.array(state, dtype=np.float64, copy=True)
    a = np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0)
    x, y, z = s[:, 0].copy(), s[:, 1].copy(), s[:, 2].copy()
    qw, qx, qy, qz = s[:, 3].copy(), s[:, 4].copy(), s[:, 5].copy(), s[:, 6].copy()
    j = s[:, 7:15].copy()
    vx, vy, vz = s[:, 15].copy(), s[:, 16].copy(), s[:, 17].copy()
    wx, wy, wz = s[:, 18].copy(), s[:, 19].copy(), s[:, 20].copy()
    jd = s[:, 21:29].copy()
    h = m.h
    for _ in range(int(m.frame_skip)):
        fxb = np.zeros_like(x)
        fyb = np.zeros_like(x)
        tz = np.zeros_like(x)
        tx = np.zeros_like(x)
        ty = np.zeros_like(x)
        lift_sum = np.zeros_like(x)
        for l in range(4):
            hip, ank = j[:, 2 * l], j[:, 2 * l + 1]
            hd, ad = jd[:, 2 * l], jd[:, 2 * l + 1]
            over = np.maximum(hip - m.hip_lim, 0.0) - np.maximum(-m.hip_lim - hip, 0.0)
            hdd = m.k_act * a[:, 2 * l] - m.k_spr * hip - m.k_dmp * hd - m.k_lim * over
            over = np.maximum(ank - m.ank_hi, 0.0) - np.maximum(m.ank_lo - ank, 0.0)
            add = m.k_act * a[:, 2 * l + 1] - m.k_spr * (ank - m.ank_rest) - m.k_dmp * ad - m.k_lim * over
            hd = hd + h * hdd
            ad = ad + h * add
            hip = hip + h * hd
            ank = ank + h * ad
            j[:, 2 * l], j[:, 2 * l + 1] = hip, ank
            jd[:, 2 * l], jd[:, 2 * l + 1] = hd, ad
            c = 0.5 * (1.0 + np.tanh(m.contact_gain * (ank - m.ank_rest)))
            push = -(m.leg_r * hd) * c                       # the loaded foot sweeps back, the torso goes the other way
            fxb = fxb + (-m.sphi[l]) * push
            fyb = fyb + m.cphi[l] * push
            tz = tz + m.leg_r * push
            lift = c * (ank - m.ank_rest)
            lift_sum = lift_sum + lift
            tx = tx + m.sphi[l] * (m.k_lift * lift)
            ty = ty - m.cphi[l] * (m.k_lift * lift)
        yaw = np.arctan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz))
        cy, sy = np.cos(yaw), np.sin(yaw)
        axw = m.k_push * (cy * fxb - sy * fyb) - m.c_lin * vx
        ayw = m.k_push * (sy * fxb + cy * fyb) - m.c_lin * vy
        vx = vx + h * axw
        vy = vy + h * ayw
        x = x + h * vx
        y = y + h * vy
        z_ref = m.z0 + m.z_gain * (0.25 * lift_sum)
        vz = vz + h * (m.k_z * (z_ref - z) - m.c_z * vz)
        z = z + h * vz
        # world up in the body frame (third row of R); e_z x u = (-u1, u0, 0) turns the body z axis towards it
        u0 = 2.0 * (qx * qz - qw * qy)
        u1 = 2.0 * (qy * qz + qw * qx)
        wx = wx + h * (tx - m.c_ang * wx + m.k_up * (-u1))
        wy = wy + h * (ty - m.c_ang * wy + m.k_up * u0)
        wz = wz + h * (m.k_yaw * tz - m.c_ang * wz)
        hw = 0.5 * h
        nqw = qw + hw * (-(qx * wx) - qy * wy - qz * wz)
        nqx = qx + hw * (qw * wx + qy * wz - qz * wy)
        nqy = qy + hw * (qw * wy + qz * wx - qx * wz)
        nqz = qz + hw * (qw * wz + qx * wy - qy * wx)
        inv = 1.0 / np.sqrt(nqw * nqw + nqx * nqx + nqy * nqy + nqz * nqz)
        qw, qx, qy, qz = nqw * inv, nqx * inv, nqy * inv, nqz * inv
    out = np.empty_like(s)
    out[:, 0], out[:, 1], out[:, 2] = x, y, z
    out[:, 3], out[:, 4], out[:, 5], out[:, 6] = qw, qx, qy, qz
    out[:, 7:15] = j
    out[:, 15], out[:, 16], out[:, 17] = vx, vy, vz
    out[:, 18], out[:, 19], out[:, 20] = wx, wy, wz
    out[:, 21:29] = jd
    return out


# --------------------------------------------------------------------------- chunk rollout
def ant_rollout_chunk(state, actions, step_fn, maze, desired_xy, s_global, A=None, ball_radius=ANT_BALL_RADIUS):
    """planners/base_planner.py:257-320 for the ant, batched: A env steps, after each the goal test (:296-297) and
    ``check_collision`` (:154-155,306); collision wins over goal (:306 before :314); on goal the remaining actions are zeroed
    (:315) and the remaining state rows stay zero (:282).

    state (B, 29), actions (B, >= A, 8); ``step_fn(i, cur (n, 29), act (n, 8), rows (n,)) -> (n, 29)`` is the env step of
    rows ``rows`` (the tape or the surrogate).  Same return dict as oracle.geometry.rollout_chunk."""
    state = np.asarray(state, dtype=np.float64)
    actions = np.array(actions, dtype=np.float64)
    B = state.shape[0]
    A = actions.shape[1] if A is None else A
    actions = actions[:, :A].copy()
    states = np.zeros((B, A + 1, S_DIM))
    states[:, 0] = state
    cur = state.copy()
    status = np.zeros(B, dtype=np.int32)
    n_steps = np.zeros(B, dtype=np.int32)
    alive = np.ones(B, dtype=bool)
    for i in range(A):
        rows = np.nonzero(alive)[0]
        if rows.size == 0:
            break
        cur[rows] = step_fn(i, cur[rows], actions[rows, i], rows)
        states[rows, i + 1] = cur[rows]
        n_steps[rows] = i + 1
        done = np.zeros(B, dtype=bool)
        coll = np.zeros(B, dtype=bool)
        done[rows] = ant_goal_reached(cur[rows], desired_xy, s_global)
        coll[rows] = is_colliding_ant(cur[rows], maze, ball_radius, s_global)
        status[coll] = G.STATUS_COLLIDED
        goal_only = done & ~coll
        status[goal_only] = G.STATUS_GOAL
        for b in np.nonzero(goal_only)[0]:
            actions[b, i + 1:] = 0.0
        alive &= ~(coll | done)
    return dict(end_state=cur, status=status, n_steps=n_steps, states=states, actions=actions)


# --------------------------------------------------------------------------- sampling
def draw_candidate_ant(tape, width, length, s_global, goal_state, goal_sample_rate=0.15, goal_conditioning_bias=0.85):
    """planners/base_planner.py:162-163,193-207 + planners/RRT.py:153-156 on a RandomTape (oracle/rrt.py): python
    ``random`` for the two coins, two np.random uniforms for a non-goal sample."""
    if tape.py.random() > goal_sample_rate:
        sample = np.zeros(S_DIM)
        x = tape.np.uniform(-s_global * width / 2, s_global * width / 2, size=(1, 1))
        y = tape.np.uniform(-s_global * length / 2, s_global * length / 2, size=(1, 1))
        sample[0], sample[1] = x[0, 0], y[0, 0]
    else:
        sample = np.array(goal_state, dtype=np.float64).copy()
    if tape.py.random() > goal_conditioning_bias:
        cond = sample[:2].copy()
    else:
        cond = np.asarray(goal_state, dtype=np.float64)[:2].copy()
    return sample, cond


# --------------------------------------------------------------------------- round-based planner (29-d)
class OracleAntPlanner:
    """oracle.rrt.OraclePlanner for the ant: rounds of B candidates against the tree snapshot, accepted in candidate order.

    ``sampler(cand_idx, chunk, hist (n, h, 29), prev_action (n, 8), has_prev (n,), cond_goal (n, 2), local_map (n, N, N))
    -> (n, >= A, 8)`` actions; ``step_fn(cand_idx (n,), chunk, i, cur (n, 29), act (n, 8)) -> (n, 29)`` the env step (global
    candidate indices: the tape's key).  B = 1 is planners/RRT.py:131-219 step for step (run_type 0)."""

    def __init__(self, maze, start_state, goal_state, desired_goal_xy, sampler, step_fn, edge_length=48, action_horizon=2,
                 local_map_size=16, local_map_scale=0.8, s_global=4.0, goal_sample_rate=0.15, goal_conditioning_bias=0.85,
                 ball_radius=ANT_BALL_RADIUS, obs_history=3):
        self.maze = np.asarray(maze, dtype=np.float32)
        self.start_state = np.asarray(start_state, dtype=np.float64)
        self.goal_state = np.asarray(goal_state, dtype=np.float64)
        self.desired = np.asarray(desired_goal_xy, dtype=np.float64)
        self.sampler, self.step_fn = sampler, step_fn
        self.H, self.A = edge_length, action_horizon
        self.n_chunks = edge_length // action_horizon
        self.lm_size, self.lm_scale, self.s_global = local_map_size, local_map_scale, s_global
        self.center = G.map_center(self.maze, s_global)       # base_planner.py:87-88 via maze_data (already scaled)
        self.gsr, self.gcb = goal_sample_rate, goal_conditioning_bias
        self.ball_radius, self.obs_history = ball_radius, obs_history
        self.states = [self.start_state.copy()]
        self.parents = [-1]
        self.last_action = [np.zeros(A_DIM)]
        self.has_prev = [False]
        self.edge_states = [None]
        self.edge_actions = [None]
        self.iterations = 0
        self.candidates = 0
        self.goal_node = None

    def __len__(self):
        return len(self.states)

    def node_history(self, n):
        """planners/RRT.py:146-147: the states the first sampler call of a child sees."""
        if self.edge_states[n] is None:
            return self.states[n][None, :]
        return self.edge_states[n]

    def expand_round(self, samples, cond_goals):
        samples = np.asarray(samples, dtype=np.float64)
        cond_goals = np.asarray(cond_goals, dtype=np.float64)
        B = samples.shape[0]
        A, nC, h = self.A, self.n_chunks, self.obs_history
        xy = np.asarray(self.states)[:, :2]
        parent = G.nn_argmin(samples[:, :2], xy)
        cur = np.asarray(self.states)[parent].copy()
        prev_a = np.asarray(self.last_action)[parent].copy()
        has_prev = np.asarray(self.has_prev)[parent].copy()
        hist = np.zeros((B, h, S_DIM))
        hist_n = np.zeros(B, dtype=np.int32)
        for b in range(B):
            hs = self.node_history(int(parent[b]))[-h:]
            hist_n[b] = len(hs)
            hist[b, h - len(hs):] = hs
        cand_idx = np.arange(self.candidates, self.candidates + B)
        all_states = np.zeros((B, nC, A + 1, S_DIM))
        all_actions = np.zeros((B, nC, A, A_DIM))
        chunk_steps = np.zeros((B, nC), dtype=np.int32)
        chunks_run = np.zeros(B, dtype=np.int32)
        final_status = np.zeros(B, dtype=np.int32)
        alive = np.ones(B, dtype=bool)
        conds = {}
        for jc in range(nC):
            idx = np.nonzero(alive)[0]
            if idx.size == 0:
                break
            lm = G.create_local_map(self.maze, cur[idx, 0], cur[idx, 1], cur[idx, 2], self.lm_size, self.lm_scale,
                                    self.s_global, self.center)     # RRT.py:158-166: "yaw" = element 2 (the torso height)
            acts = np.zeros((idx.size, A, A_DIM))
            # the sampler sees ragged histories (1 row at the root, 3 afterwards): group by length
            for n in np.unique(hist_n[idx]):
                sel = idx[hist_n[idx] == n]
                out = self.sampler(cand_idx[sel], jc, hist[sel, h - n:], prev_a[sel], has_prev[sel], cond_goals[sel],
                                   lm[hist_n[idx] == n])
                acts[hist_n[idx] == n] = np.asarray(out, dtype=np.float64)[:, :A]
            r = ant_rollout_chunk(cur[idx], acts, lambda i, c, a, rows, jc=jc, idx=idx: self.step_fn(cand_idx[idx[rows]], jc, i, c, a),
                                  self.maze, self.desired, self.s_global, A, self.ball_radius)
            all_states[idx, jc] = r["states"]
            all_actions[idx, jc] = r["actions"]
            chunk_steps[idx, jc] = r["n_steps"]
            chunks_run[idx] += 1
            cur[idx] = r["end_state"]
            final_status[idx] = r["status"]
            ok = r["status"] == G.STATUS_OK
            prev_a[idx[ok]] = r["actions"][ok, A - 1]                 # RRT.py:188 prev_actions = curr_action_seq
            has_prev[idx[ok]] = True
            n_new = min(h, A + 1)                                     # RRT.py:190 prev_states = curr_states_seq
            hist[idx[ok]] = 0.0
            hist[idx[ok], h - n_new:] = r["states"][ok][:, -n_new:]
            hist_n[idx[ok]] = n_new
            alive[idx] = ok
        accepted = []
        for b in range(B):
            self.candidates += 1
            self.iterations += int(chunks_run[b])
            if final_status[b] == G.STATUS_COLLIDED:
                continue
            es = np.concatenate([all_states[b, jc] for jc in range(chunks_run[b])])
            ea = np.concatenate([all_actions[b, jc] for jc in range(chunks_run[b])])
            ea = ea[~(ea == 0).all(axis=1)]                            # RRT.py:196-197
            es = es[~(es == 0).all(axis=1)]                            # RRT.py:198-199
            self.states.append(cur[b].copy())
            self.parents.append(int(parent[b]))
            self.edge_states.append(es)
            self.edge_actions.append(ea)
            self.last_action.append(ea[-1].copy() if len(ea) else np.zeros(A_DIM))
            self.has_prev.append(len(ea) > 0)
            accepted.append(len(self.states) - 1)
            if final_status[b] == G.STATUS_GOAL:
                self.goal_node = accepted[-1]
                break
        return dict(parent=parent, status=final_status, end_state=cur, accepted=accepted, chunks_run=chunks_run,
                    chunk_steps=chunk_steps, states=all_states, actions=all_actions)

    def fallback_node(self):
        """planners/RRT.py:233-237 (run_type 0): nearest node to goal_state xy among nodes 1.."""
        if len(self.states) < 2:
            return None
        d = np.asarray(self.states)[1:, :2] - self.goal_state[:2]
        return 1 + int(np.argmin(G.norm2(d[:, 0], d[:, 1])))

    def path_to(self, node):
        """planners/base_planner.py:342-363."""
        path, actions = [], []
        n = node
        while n != -1:
            seg = [self.states[n]]
            if self.edge_states[n] is not None:
                seg = list(self.edge_states[n]) + seg
            path = seg + path
            if self.edge_actions[n] is not None:
                actions = list(self.edge_actions[n]) + actions
            n = self.parents[n]
        return (np.array(path, dtype=np.float32) if path else None,
                np.array(actions, dtype=np.float32) if actions else None)

    def plan(self, tape, n_candidates, batch=1):
        H, W = self.maze.shape
        while self.goal_node is None and self.candidates < n_candidates:
            B = min(batch, n_candidates - self.candidates)
            s, c = np.zeros((B, S_DIM)), np.zeros((B, 2))
            for i in range(B):
                s[i], c[i] = draw_candidate_ant(tape, W, H, self.s_global, self.goal_state, self.gsr, self.gcb)
            self.expand_round(s, c)
        reached = self.goal_node is not None
        node = self.goal_node if reached else self.fallback_node()
        path, actions = self.path_to(node) if node is not None else (None, None)
        return reached, path, actions


# --------------------------------------------------------------------------- observation tapes
class AntObsTape:
    """Next observations as a pure function of (seed, global candidate, chunk, step): per candidate a smooth walk that starts
    in a free cell of the maze, drifts a few tenths of a cell per env step, keeps a plausible torso height and a unit
    quaternion that is upright for most candidates and tips over for a few (the upside-down test), and -- for every
    ``goal_every``-th candidate -- bends towards ``desired_xy`` so that some edges reach the goal radius.  It does NOT follow
    the actions: it stands in for the physics only as far as the planner's bookkeeping is concerned."""

    def __init__(self, seed, maze, s_global, n_chunks, A, desired_xy=None, goal_every=7, step=0.35):
        self.seed, self.maze, self.s = int(seed), np.asarray(maze), float(s_global)
        self.nC, self.A, self.desired, self.goal_every, self.step = n_chunks, A, desired_xy, goal_every, step
        self._cache = {}

    def _walk(self, c):
        c = int(c)
        if c in self._cache:
            return self._cache[c]
        rng = np.random.default_rng([self.seed, c])
        H, W = self.maze.shape
        free = np.argwhere(self.maze[1:-1, 1:-1] == 0) + 1
        cell = free[rng.integers(0, len(free))]
        n = self.nC * self.A
        xy0 = np.array([((cell[1] + 0.5) - W / 2) * self.s, (H / 2 - (cell[0] + 0.5)) * self.s]) + rng.uniform(-0.5, 0.5, 2)
        heading = rng.uniform(-np.pi, np.pi) + np.cumsum(rng.normal(0.0, 0.25, n))
        d = self.step * np.stack([np.cos(heading), np.sin(heading)], axis=1)
        xy = xy0 + np.cumsum(d, axis=0)
        if self.desired is not None and self.goal_every and c % self.goal_every == 3:
            k = int(rng.integers(n // 3, n))
            w = np.clip(np.arange(1, n + 1) / k, 0.0, 1.0)[:, None]
            xy = (1 - w) * xy + w * (np.asarray(self.desired) + rng.uniform(-0.4, 0.4, 2))
        o = rng.normal(0.0, 0.6, (n, S_DIM))
        o[:, 0:2] = xy
        o[:, 2] = rng.uniform(0.45, 0.8, n)
        tip = rng.random() < 0.06                              # a few candidates roll over somewhere along the edge
        ang = np.cumsum(rng.normal(0.0, 0.12 if not tip else 0.45, n))
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        o[:, 3] = np.cos(ang / 2)
        o[:, 4:7] = np.sin(ang / 2)[:, None] * axis
        self._cache[c] = o.reshape(self.nC, self.A, S_DIM)
        return self._cache[c]

    def rows(self, cand_idx):
        """(n,) global candidate indices -> (n, n_chunks, A, 29)."""
        return np.stack([self._walk(c) for c in np.asarray(cand_idx).reshape(-1)])

    def step_fn(self):
        def fn(cand_idx, chunk, i, cur, act):
            return np.stack([self._walk(c)[chunk, i] for c in cand_idx])
        return fn


class AntActionTape:
    """(seed, candidate, chunk) -> (P, 8) actions in [-1.2, 1.2] (the env clips to [-1, 1]); stands in for the denoiser."""

    def __init__(self, seed, P=16):
        self.seed, self.P = int(seed), P

    def actions(self, cand_idx, chunk):
        out = np.empty((len(cand_idx), self.P, A_DIM))
        for k, c in enumerate(np.asarray(cand_idx).reshape(-1)):
            rng = np.random.default_rng([self.seed, int(c), int(chunk)])
            base = rng.uniform(-1.0, 1.0, A_DIM)
            out[k] = np.clip(base + rng.normal(0.0, 0.3, (self.P, A_DIM)), -1.2, 1.2)
        return out

    def sampler(self):
        def fn(cand_idx, chunk, hist, prev_action, has_prev, cond_goal, local_map):
            return self.actions(cand_idx, chunk)
        return fn
