"""Numpy (float64) restatement of the reference's per-candidate geometry.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All functions are batched
over a leading candidate axis but keep the reference's per-element operation
order, so B = 1 reproduces the reference call exactly.

Reference sites (relative to /root/reference):
  car dynamics      car_env.py:356-396 (step), :341-354 (goal), :586-597 (bounds)
  cell <-> xy       car_env.py:189-201
  collision         common/map_utils.py:103-115 (two balls), :221-329 (grid test)
  local map         common/map_utils.py:391-459
  lidar             lidar_sim/lidar_2d_sim.py:14-16, :18-98
  propagate chunk   planners/base_planner.py:257-320
  obstacle ahead    planners/RRT.py:61-81
  nearest node      planners/RRT.py:49-51 (scipy KDTree, k = 1, first two dims)
"""
from __future__ import annotations

import ctypes

import numpy as np

# --------------------------------------------------------------------------- constants
# car_env.py:32,47-53
CAR_DT = 1.0 / 50.0
CAR_M = 0.043
CAR_C1 = 0.5
CAR_C2 = 15.5
CAR_CM1 = 0.28
CAR_CM2 = 0.05
CAR_CR0 = 0.011
CAR_CR2 = 0.006
# car_env.py:56-66 with bicycle_model() bounds :590-597 (float32 Box, values exact)
ACT_LOW = np.array([-10.0, -2.0])
ACT_HIGH = np.array([10.0, 2.0])
GOAL_RADIUS = 0.5          # car_env.py:350
BALL_RADIUS = 0.1          # map_utils.py:103
CAR_LENGTH = 0.15          # map_utils.py:103

STATUS_OK = 0              # chunk finished, neither goal nor collision
STATUS_GOAL = 1            # `done is True`  (base_planner.py:300,314-317)
STATUS_COLLIDED = 2        # `done is None`  (base_planner.py:306-312)


# --------------------------------------------------------------------------- cell <-> xy
def map_center(maze: np.ndarray, scale: float = 1.0):
    """car_env.py:84-87: x_center = W/2*s, y_center = H/2*s."""
    return maze.shape[1] / 2 * scale, maze.shape[0] / 2 * scale


def cell_rowcol_to_xy(rowcol, maze, scale: float = 1.0):
    """car_env.py:189-194."""
    xc, yc = map_center(maze, scale)
    rc = np.asarray(rowcol, dtype=np.float64)
    x = (rc[..., 1] + 0.5) * scale - xc
    y = yc - (rc[..., 0] + 0.5) * scale
    return np.stack([x, y], axis=-1)


def cell_xy_to_rowcol(xy, maze, scale: float = 1.0, floor_enable: bool = True):
    """car_env.py:196-201."""
    xc, yc = map_center(maze, scale)
    xy = np.asarray(xy, dtype=np.float64)
    i = (yc - xy[..., 1]) / scale
    j = (xy[..., 0] + xc) / scale
    ret = np.stack([i, j], axis=-1)
    return np.floor(ret) if floor_enable else ret


# --------------------------------------------------------------------------- dynamics
def car_step(state: np.ndarray, action: np.ndarray) -> np.ndarray:
    """One explicit-Euler step of the bicycle model, car_env.py:356-396.

    state (..., 6) = x, y, psi, v, D, delta; action (..., 2) = dD, ddelta.
    """
    state = np.asarray(state, dtype=np.float64)
    a = np.clip(np.asarray(action, dtype=np.float64), ACT_LOW, ACT_HIGH)   # :371
    psi = state[..., 2]
    v = state[..., 3]
    D = state[..., 4]
    delta = state[..., 5]
    Fxd = (CAR_CM1 - CAR_CM2 * v) * D - CAR_CR2 * (v * v) - CAR_CR0 * np.tanh(5.0 * v)   # :380
    dot = np.stack([
        v * np.cos(psi + CAR_C1 * delta),
        v * np.sin(psi + CAR_C1 * delta),
        v * CAR_C2 * delta,
        (Fxd / CAR_M) * np.cos(CAR_C1 * delta),
        a[..., 0],
        a[..., 1],
    ], axis=-1)
    return state + CAR_DT * dot                                                           # :390


_libm = ctypes.CDLL("libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3
_fma = np.frompyfunc(_libm.fma, 3, 1)


def norm2(dx, dy):
    """np.linalg.norm of a 2-vector as numpy evaluates it (1-D path: sqrt(x.dot(x)), and the
    BLAS ddot of two elements rounds as fma(x1, x1, x0*x0) -- checked bit for bit against
    np.linalg.norm on 2e5 random vectors in the build container)."""
    dx = np.asarray(dx, dtype=np.float64)
    dy = np.asarray(dy, dtype=np.float64)
    return np.sqrt(np.asarray(_fma(dy, dy, dx * dx), dtype=np.float64))


def goal_reached(state: np.ndarray, goal_xy: np.ndarray) -> np.ndarray:
    """car_env.py:341-354: np.linalg.norm(xy - goal) < 0.5."""
    d = np.asarray(state)[..., :2] - np.asarray(goal_xy)
    return norm2(d[..., 0], d[..., 1]) < GOAL_RADIUS


# --------------------------------------------------------------------------- collision
def _ball_collides(x, y, maze, scale=1.0, r=BALL_RADIUS):
    """common/map_utils.py:221-329 for one ball per row of x/y (any leading shape).

    The reference's early returns only ever return arrays that are already True
    where the final OR would be True, and an out-of-bounds ball short-circuits the
    whole call (:258-259); both collapse to the OR below once ``.any()`` over the two
    balls is applied (map_utils.py:115).  The column clip of the corner test uses
    ``map_length`` (rows), :326 -- reproduced.
    """
    H, W = maze.shape
    xc = W / 2 * scale
    yc = H / 2 * scale
    rows = np.floor((yc - y) / scale).astype(np.int64)
    cols = np.floor((x + xc) / scale).astype(np.int64)
    oob = (rows < 0) | (rows >= H) | (cols < 0) | (cols >= W)
    rs = np.clip(rows, 0, H - 1)
    cs = np.clip(cols, 0, W - 1)
    coll = oob | (maze[rs, cs] == 1)
    cell_x = (cs + 0.5) * scale - xc
    cell_y = yc - (rs + 0.5) * scale
    half = scale / 2
    x_min, x_max = cell_x - half, cell_x + half
    y_min, y_max = cell_y - half, cell_y + half
    right, left, top, bottom = x + r, x - r, y + r, y - r
    coll |= (right > x_max) & (maze[rs, np.clip(cs + 1, 0, W - 1)] == 1)
    coll |= (left < x_min) & (maze[rs, np.clip(cs - 1, 0, W - 1)] == 1)
    coll |= (top > y_max) & (maze[np.clip(rs - 1, 0, H - 1), cs] == 1)
    coll |= (bottom < y_min) & (maze[np.clip(rs + 1, 0, H - 1), cs] == 1)
    for cx, cy, ci, cj in ((x_max, y_max, rs - 1, cs + 1), (x_min, y_max, rs - 1, cs - 1),
                           (x_max, y_min, rs + 1, cs + 1), (x_min, y_min, rs + 1, cs - 1)):
        dist = np.hypot(cx - x, cy - y)
        invalid = (ci < 0) | (ci >= H) | (cj < 0) | (cj >= W)
        ci2 = np.clip(ci, 0, H - 1)
        cj2 = np.clip(cj, 0, H - 1)              # sic: map_length, map_utils.py:326
        # W < H would raise IndexError in the reference; every shipped maze has H <= W.
        cj2 = np.minimum(cj2, W - 1)
        coll |= invalid | ((dist < r) & (maze[ci2, cj2] == 1))
    return coll


def is_colliding_car(state: np.ndarray, maze: np.ndarray) -> np.ndarray:
    """common/map_utils.py:103-115: two balls +-0.075 m along the heading."""
    state = np.asarray(state, dtype=np.float64)
    ox = (CAR_LENGTH * 0.5) * np.cos(state[..., 2])
    oy = (CAR_LENGTH * 0.5) * np.sin(state[..., 2])
    x, y = state[..., 0], state[..., 1]
    front = _ball_collides(x + ox, y + oy, maze)
    rear = _ball_collides(x - ox, y - oy, maze)
    return front | rear


# --------------------------------------------------------------------------- chunk rollout
def rollout_chunk(state, actions, maze, goal_xy, action_horizon=None):
    """planners/base_planner.py:257-320 for a batch of candidates.

    state (B, 6), actions (B, n, 2) with n >= action_horizon.
    Returns dict with
      end_state (B, 6)      the reference's ``obs``
      status    (B,) int    STATUS_OK / STATUS_GOAL / STATUS_COLLIDED
      n_steps   (B,) int    env steps executed (collision step included)
      states    (B, A+1, 6) the reference's ``states_sequence`` BEFORE its final
                            slicing: rows > n_steps stay zero (:282)
      actions   (B, A, 2)   the *unclipped* input actions with rows after the goal
                            step zeroed (:314-317)
      goal_at_collision (B,) the step that collided was also inside the goal radius.
                            In the reference this leaves ``env.done`` latched
                            (car_env.py:254,266) -- see oracle/rrt.py "sticky done".
    A collided candidate's reference return values are ``actions[:i]`` /
    ``states[:i]`` with i = n_steps - 1 (:309-312).
    """
    state = np.asarray(state, dtype=np.float64)
    actions = np.array(actions, dtype=np.float64)
    B = state.shape[0]
    A = actions.shape[1] if action_horizon is None else action_horizon
    actions = actions[:, :A].copy()
    states = np.zeros((B, A + 1, 6))
    states[:, 0] = state
    cur = state.copy()
    status = np.zeros(B, dtype=np.int32)
    n_steps = np.zeros(B, dtype=np.int32)
    alive = np.ones(B, dtype=bool)
    goal_at_collision = np.zeros(B, dtype=bool)
    for i in range(A):
        if not alive.any():
            break
        nxt = car_step(cur, actions[:, i])
        cur = np.where(alive[:, None], nxt, cur)
        states[alive, i + 1] = cur[alive]
        n_steps[alive] = i + 1
        done = goal_reached(cur, goal_xy) & alive
        coll = is_colliding_car(cur, maze) & alive
        status[coll] = STATUS_COLLIDED                     # collision wins (:306 before :314)
        goal_at_collision |= coll & done
        goal_only = done & ~coll
        status[goal_only] = STATUS_GOAL
        for b in np.nonzero(goal_only)[0]:
            actions[b, i + 1:] = 0.0
        alive &= ~(coll | done)
    return dict(end_state=cur, status=status, n_steps=n_steps, states=states, actions=actions,
                goal_at_collision=goal_at_collision)


# --------------------------------------------------------------------------- local map
def local_axis(n: int, scale: float) -> np.ndarray:
    """common/map_utils.py:422-423."""
    L = n * scale
    return np.linspace(-L / 2 + scale / 2, L / 2 - scale / 2, n)


def create_local_map(maze, x, y, theta, n, scale, s_global, center):
    """common/map_utils.py:391-459 -> (B, n, n) float32 in {0, 1}."""
    maze = np.asarray(maze, dtype=np.float32)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    y = np.atleast_1d(np.asarray(y, dtype=np.float64))
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    xs = local_axis(n, scale)
    xl, yl = np.meshgrid(xs, xs)
    xl = xl.reshape(1, -1)
    yl = yl.reshape(1, -1)
    c = np.cos(theta)[:, None]
    s = np.sin(theta)[:, None]
    xg = c * xl - s * yl + x[:, None]
    yg = s * xl + c * yl + y[:, None]
    yi = np.floor((center[1] - yg) / s_global).astype(np.int64)
    xi = np.floor((xg + center[0]) / s_global).astype(np.int64)
    xi = np.clip(xi, 0, maze.shape[1] - 1)
    yi = np.clip(yi, 0, maze.shape[0] - 1)
    return maze[yi, xi].reshape(len(x), n, n)


# --------------------------------------------------------------------------- nearest node
def nn_argmin(queries_xy: np.ndarray, nodes_xy: np.ndarray) -> np.ndarray:
    """planners/RRT.py:49-51 (KDTree.query k=1 over the first two state dims).

    Brute force over squared distances; ties resolve to the lowest node index.
    """
    q = np.asarray(queries_xy, dtype=np.float64)
    n = np.asarray(nodes_xy, dtype=np.float64)
    dx = q[:, None, 0] - n[None, :, 0]
    dy = q[:, None, 1] - n[None, :, 1]
    return np.argmin(dx * dx + dy * dy, axis=1).astype(np.int32)


# --------------------------------------------------------------------------- obstacle ahead
def check_obstacle_ahead(state, maze):
    """planners/RRT.py:61-81 (only used when run_type > 0)."""
    state = np.atleast_2d(np.asarray(state, dtype=np.float64))
    rc = cell_xy_to_rowcol(state[:, :2], maze, floor_enable=False)
    t = np.linspace(0, 1.5, 30)
    theta = state[:, 2]
    px = t[None, :] * np.cos(-theta)[:, None] + rc[:, 1][:, None]
    py = t[None, :] * np.sin(-theta)[:, None] + rc[:, 0][:, None]
    qx = np.clip(px.astype(np.int64), 0, maze.shape[1] - 1)
    qy = np.clip(py.astype(np.int64), 0, maze.shape[0] - 1)
    return np.any(maze[qy, qx] != 0, axis=1)


# --------------------------------------------------------------------------- lidar
LIDAR_ANGLES_DEG = np.arange(-360 / 2, 360 / 2 + 2.0, 2.0)      # lidar_2d_sim.py:14-16 (181)


def lidar_cast_ray(pose, maze, angle_deg):
    """lidar_sim/lidar_2d_sim.py:47-98, one ray.  pose = (x_col, y_row, yaw) in cells.

    Keeps the reference quirks: yaw (radians) is added to the angle in degrees (:53),
    ``maze_width, maze_height = maze.shape`` (:51), first admissible border in the
    order Left/Right/Bottom/Top (:57-82).
    Returns (distance, endpoint(2,), visited cells (k, 2) int [x, y], hit flag).
    """
    x0, y0, yaw = pose
    mw, mh = maze.shape
    ang = np.deg2rad(yaw + angle_deg)
    rv = np.array([np.cos(ang), np.sin(ang)])
    p = np.array([x0, y0], dtype=np.float64)
    borders = [(np.array([0, 0]), np.array([0, mh])), (np.array([mw, 0]), np.array([mw, mh])),
               (np.array([0, 0]), np.array([mw, 0])), (np.array([0, mh]), np.array([mw, mh]))]
    last = None
    for b0, b1 in borders:
        di = (b1 - b0).astype(float)
        Amat = np.column_stack((rv, -di))
        try:
            t, s = np.linalg.solve(Amat, b0 - p)
        except np.linalg.LinAlgError:
            continue
        if t >= 0 and 1 >= s >= 0:
            last = t * rv + p
            break
    ts = np.arange(0, 1, step=0.1 / np.linalg.norm(last - p))
    dots = p[None] + ts[None].T * (last - p)[None]
    q = np.floor(dots).astype(int)
    q = np.clip(q, [0, 0], [mw - 1, mh - 1])
    occ = maze[q.T[1], q.T[0]]
    hit = bool(np.any(occ == 1))
    if hit:
        first = int(np.where(occ == 1)[0][0])
        obstacle = dots[first]
    else:
        first = len(q)
        obstacle = last
    return float(np.linalg.norm(obstacle - p)), obstacle, q[:first], hit


def lidar_scan(pose, maze):
    """lidar_sim/lidar_2d_sim.py:18-45 with noise_std = 0 (the default, :6).

    Returns distances (181,), endpoints (181, 2), visited (k, 2), hits (181,) bool.
    The endpoint is re-derived from the distance as in :30-40.
    """
    pose = np.asarray(pose, dtype=np.float64)
    yaw = pose[-1]
    dists, ends, visited, hits = [], [], [], []
    for ang in LIDAR_ANGLES_DEG:
        d, _, cells, hit = lidar_cast_ray(pose, maze, ang)
        visited.extend(cells)
        a = np.deg2rad(yaw + ang)
        d = float(np.clip(d, 0, 300))
        dists.append(d)
        ends.append((pose[0] + d * np.cos(a), pose[1] + d * np.sin(a)))
        hits.append(hit)
    vis = np.array(visited, dtype=np.int64).reshape(-1, 2)
    return np.array(dists), np.array(ends), vis, np.array(hits)
