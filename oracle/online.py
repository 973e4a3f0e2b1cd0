"""TEST INFRASTRUCTURE -- CPU restatement of the online re-planning driver's sensing / plan-following
steps (reference: run_scenarios_with_lidar_DiTree.py:112-127 scan_and_update_maze, :158-181
check_no_obstacles_in_path, :470-506 the action-execution loop).  Only tests/, bench.py's cpu_baseline
and __graft_entry__.smoke() may import this package; the product never does.

Pinned by tests/golden/online.npz: the two reference functions are taken from the script's text at
generation time (the script itself cannot be imported: playsound, minari, gymnasium ... are absent) and
run against the reference's Lidar2DSim; the execution loop is driven through the reference planner's
``propagate_action_sequence_env`` (tests/golden/make_golden.py, gen_online).
"""
from __future__ import annotations

import numpy as np

from . import geometry as G

EV_ACTIONS_DONE, EV_GOAL, EV_COLLISION, EV_OBSTACLE = 0, 1, 2, 3


def lidar_pose(state, maze):
    """:113-116: (x, y, psi) -> (col, row, psi) in fractional cell units; the yaw stays in radians."""
    rc = G.cell_xy_to_rowcol(np.asarray(state, dtype=np.float64)[:2], maze, floor_enable=False)
    return np.array([rc[1], rc[0], float(state[2])])


def scan_and_update_maze(state, maze_known, maze_true, scanned):
    """:112-127.  Scans the TRUE maze from the robot pose and writes the ray end cells as occupied into
    the known maze and the scanned maze (visited cells = 2, end cells = 1); both are updated in place."""
    pose = lidar_pose(state, maze_known)
    _, ends, visited, _ = G.lidar_scan(pose, maze_true)
    cells = np.floor(ends).astype(int)
    maze_known[cells[:, 1], cells[:, 0]] = 1
    if len(visited):
        scanned[visited[:, 1], visited[:, 0]] = 2
    scanned[cells[:, 1], cells[:, 0]] = 1
    return pose


def check_no_obstacles_in_path(scanned, path_xy):
    """:158-181: index of the first path point whose cell is marked occupied in the scanned maze, else -1."""
    H, W = scanned.shape
    p = np.asarray(path_xy)[:, :2]
    # cell_xy_to_rowcol(x, floor_enable=False) per point, in the dtype of the path (float32 paths stay float32)
    row = (np.asarray(H / 2, dtype=p.dtype) - p[:, 1]) / np.asarray(1.0, dtype=p.dtype)
    col = (p[:, 0] + np.asarray(W / 2, dtype=p.dtype)) / np.asarray(1.0, dtype=p.dtype)
    qx, qy = np.floor(col).astype(int), np.floor(row).astype(int)
    for i in range(len(p)):
        if scanned[qy[i], qx[i]] == 1:
            return i
    return -1


def follow_plan(state, actions, action_idx, path_xy, maze_known, maze_true, scanned, goal_xy, dt=0.02, scan_time=0.2):
    """:470-506 for run_type < 4: execute ``actions[action_idx:]`` one env step at a time
    (propagate_action_sequence_env: step, goal test, collision test on the KNOWN maze), scanning every
    time the accumulated step time exceeds ``scan_time`` and stopping at the first event.

    Returns dict(state, action_idx, executed (k, 6), event, obstacle_idx); the mazes are updated in place."""
    state = np.asarray(state, dtype=np.float64).copy()
    actions = np.asarray(actions)
    executed = []
    obstacle = -1
    t_acc = 0.0
    event = EV_ACTIONS_DONE
    while action_idx < len(actions) and obstacle < 0:
        r = G.rollout_chunk(state[None], np.asarray(actions[action_idx], dtype=np.float64)[None, None], maze_known,
                            goal_xy, action_horizon=1)
        if r["status"][0] == G.STATUS_COLLIDED:
            event = EV_COLLISION
            break
        state = r["end_state"][0].copy()
        executed.append(state.copy())
        action_idx += 1
        t_acc += dt
        if t_acc > scan_time:
            scan_and_update_maze(state, maze_known, maze_true, scanned)
            obstacle = check_no_obstacles_in_path(scanned, path_xy)
            t_acc = 0
        if r["status"][0] == G.STATUS_GOAL:
            event = EV_GOAL
            break
    if event == EV_ACTIONS_DONE and obstacle >= 0:
        event = EV_OBSTACLE
    return dict(state=state, action_idx=action_idx, executed=np.array(executed).reshape(-1, 6), event=event,
                obstacle_idx=obstacle)
