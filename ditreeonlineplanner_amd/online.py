"""Online driver steps over the HIP engine (reference: run_scenarios_with_lidar_DiTree.py:65-76 plan_path,
:112-127 scan_and_update_maze, :158-181 check_no_obstacles_in_path, :470-506 the action-execution loop).

``scan_and_update_maze`` / ``check_no_obstacles_in_path`` / ``plan_path`` keep the reference's names, arguments
and in-place numpy semantics, so the reference's driver loop runs unchanged on them.  ``follow_plan`` replaces
the driver's inner ``while`` (one Python iteration + one lidar scan every 11th step in the reference) by a single
launch of the fused kernel: dynamics, goal / collision tests, periodic scans, maze updates and the path-crossing
check stay on the GPU until the first event, and the engine's device copy of the known maze is refreshed in the
same launch (no re-upload before the next plan).
"""
from __future__ import annotations

import numpy as np
import torch

EV_ACTIONS_DONE, EV_GOAL, EV_COLLISION, EV_OBSTACLE = 0, 1, 2, 3


def plan_path(planner, curr_state, goal_state, stats2keep: dict, time_budget=None):
    """:65-76."""
    planner.plan_count += 1
    budget = planner.time_budget
    planner.reset(start_state=curr_state, goal_state=goal_state)
    if time_budget is not None:
        planner.time_budget = time_budget
    path, actions = planner.plan()
    for k in stats2keep:
        stats2keep[k] += planner.results[k]
    planner.reset(start_state=curr_state, goal_state=goal_state)
    planner.time_budget = budget
    return path, actions


def scan_and_update_maze(planner, maze_data, maze_data_with_obstacle, scanned_maze, debug=False):
    """:112-127: lidar scan of the true maze from the env state (lidar kernel), ray end cells marked occupied in
    the known and the scanned maze (both updated in place), ``planner.update_maze``."""
    state = planner.env.state
    pose = state.copy()
    pose[:2] = planner.env.cell_xy_to_rowcol(state[:2], floor_enable=False)
    pose[:2] = pose[:2][::-1]
    _, endpoints, visited = planner.env.lidar2dsim.scan(pose[:3], maze_data_with_obstacle, debug)
    ends = np.floor(endpoints).astype("int")
    maze_data[ends[:, 1], ends[:, 0]] = 1
    if len(visited):                      # (the reference cannot index an empty visited list)
        scanned_maze[visited[:, 1], visited[:, 0]] = 2
    scanned_maze[ends[:, 1], ends[:, 0]] = 1
    planner.update_maze(maze_data)


def check_no_obstacles_in_path(planner, scanned_maze, main_path_array, debug=False):
    """:158-181: first index of the path whose cell is occupied in the scanned maze, or -1."""
    rc = np.array([planner.env.cell_xy_to_rowcol(p, floor_enable=False) for p in main_path_array[:, :2]])
    q = np.floor(rc[:, ::-1]).astype("int")                    # (col, row)
    hits = np.flatnonzero(scanned_maze[q[:, 1], q[:, 0]] == 1)
    return int(hits[0]) if hits.size else -1


def follow_plan(planner, curr_state, main_actions, action_idx, main_path_array, maze_data, maze_data_with_obstacle,
                scanned_maze):
    """:470-506 (run_type < 4) as one fused launch.

    Executes ``main_actions[action_idx:]`` from ``curr_state`` until the goal is reached, the known maze is hit,
    a scan finds the planned path blocked, or the actions run out.  ``maze_data`` / ``scanned_maze`` are updated in
    place like the reference's arrays and the planner takes the new known maze.  Returns
    ``(curr_state, action_idx, executed_states (k, 6), event, obstacles_in_way)``."""
    ctx = planner.ctx
    dev = ctx.device
    if tuple(ctx.maze_shape or ()) != tuple(np.shape(maze_data)):
        ctx.upload_maze(np.asarray(maze_data, dtype=np.float32))
    st = torch.as_tensor(np.asarray(curr_state, dtype=np.float64).copy(), device=dev)
    acts = torch.as_tensor(np.ascontiguousarray(main_actions, dtype=np.float32), device=dev)
    path = torch.as_tensor(np.ascontiguousarray(np.asarray(main_path_array)[:, :2], dtype=np.float32), device=dev)
    known = torch.as_tensor(np.ascontiguousarray(maze_data, dtype=np.float32), device=dev)
    truth = torch.as_tensor(np.ascontiguousarray(maze_data_with_obstacle, dtype=np.float32), device=dev)
    scanned = torch.as_tensor(np.ascontiguousarray(scanned_maze, dtype=np.float32), device=dev)
    executed, event, nxt, obstacle = ctx.follow_plan(st, acts, int(action_idx), path, known, truth, scanned,
                                                     np.asarray(planner.env.goal, dtype=np.float64), planner.env.dt,
                                                     planner.env.lidar2dsim.scan_time)
    maze_data[...] = known.cpu().numpy()
    scanned_maze[...] = scanned.cpu().numpy()
    new_state = st.cpu().numpy()
    planner.env.set_state(new_state)
    planner.adopt_maze(maze_data)
    return new_state, nxt, executed.cpu().numpy(), event, obstacle
