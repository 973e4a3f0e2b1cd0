"""`common.map_utils` surface the reference's drivers and planners touch, on the engine.

* ``cc_calls`` -- the collision-check counter the drivers reset and read (run_scenarios.py:338,343,
  run_scenarios_with_lidar_DiTree.py:409,463; incremented by the reference in ``is_colliding_car``,
  common/map_utils.py:103-105).  ``RRT_Planner.plan`` adds the number of env steps its rounds executed on the GPU
  (one two-ball test per step, as the reference: base_planner.py:290-312).
* ``is_colliding_car`` / ``create_local_map`` / ``is_colliding_ant`` / ``is_colliding_maze`` -- same signatures, evaluated by
  the HIP kernels (bit-exact flags / maps, tests/test_gpu_geometry.py, tests/test_gpu_ant_round.py).
Anything else of the reference module (forest / PNG helpers, drone collision) is resolved lazily from the
reference checkout when one is on the path (PEP 562 ``__getattr__``)."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

cc_calls = 0


def print_calls():
    print(cc_calls)


def add_cc_calls(n: int):
    """Called by the planner facade: `n` two-ball collision tests were evaluated on the device."""
    global cc_calls
    cc_calls += int(n)


def _ctx():
    from ..ops import default_context
    return default_context()


def is_colliding_car(state, maze_map, ball_radius=0.1, car_length=0.15):
    """common/map_utils.py:103-115 on the device (a zero-velocity one-step rollout keeps the pose: its collision flag)."""
    global cc_calls
    if ball_radius != 0.1 or car_length != 0.15:
        raise NotImplementedError("the rollout kernel is built for the reference's car (two balls r = 0.1, 0.15 apart)")
    import torch
    cc_calls += 1
    ctx = _ctx()
    ctx.upload_maze(np.asarray(maze_map, dtype=np.float32), owner=None)
    st = np.zeros((1, 6))
    st[0, :3] = np.asarray(state, dtype=np.float64)[:3]
    s = torch.as_tensor(st, device=ctx.device)
    a = torch.zeros(1, 1, 2, dtype=torch.float64, device=ctx.device)
    status, _, _, _ = ctx.car_rollout(s, a, np.array([1e9, 1e9]), A=1)
    return bool((int(status.item()) & 0xFF) == 2)


def create_local_map(global_map, x, y, theta, map_size, scale, s_global, map_center):
    """common/map_utils.py:391-459: (K, N, N) occupancy windows around the poses, gathered by `local_map_kernel`."""
    import torch
    if isinstance(x, (int, float, np.generic)):
        x, y, theta = np.array([x]), np.array([y]), np.array([theta])
    x, y, theta = (np.asarray(v, dtype=np.float64).reshape(-1) for v in (x, y, theta))
    N = int(map_size) if isinstance(map_size, (int, float)) else int(map_size[0])
    gm = np.asarray(global_map)
    want = (gm.shape[1] * s_global / 2.0, gm.shape[0] * s_global / 2.0)
    if abs(map_center[0] - want[0]) > 1e-12 or abs(map_center[1] - want[1]) > 1e-12:
        raise NotImplementedError("map_center must be the centre of the global map (what every reference call passes)")
    ctx = _ctx()
    ctx.upload_maze(gm.astype(np.float32), owner=None)
    st = np.zeros((len(x), 6))
    st[:, 0], st[:, 1], st[:, 2] = x, y, theta
    out = ctx.local_map(torch.as_tensor(st, device=ctx.device), n=N, scale=float(scale), s_global=float(s_global))
    return out.cpu().numpy().astype(gm.dtype if gm.dtype.kind == "f" else np.float32)


def is_colliding_maze(state, maze_grid, maze_size_scaling=1, ball_radius=0.1):
    """common/map_utils.py:139-219 (one ball; the point-maze / ant collision test) on the device: `ant_collision` with an
    upright torso."""
    import torch
    ctx = _ctx()
    ctx.upload_maze(np.asarray(maze_grid, dtype=np.float32), owner=None)
    st = np.zeros((1, 7))
    st[0, :2] = np.asarray(state, dtype=np.float64)[:2]
    st[0, 3] = 1.0
    return bool(ctx.ant_collision(torch.as_tensor(st, device=ctx.device), float(ball_radius), float(maze_size_scaling))[0].item())


def is_colliding_ant(state, maze_map, ant_radius=1, map_scale=1):
    """common/map_utils.py:126-136: upside down (body z axis below the horizon) or the single-ball maze test, on the device."""
    import torch
    ctx = _ctx()
    ctx.upload_maze(np.asarray(maze_map, dtype=np.float32), owner=None)
    st = np.asarray(state, dtype=np.float64).reshape(1, -1)
    return bool(ctx.ant_collision(torch.as_tensor(np.ascontiguousarray(st), device=ctx.device), float(ant_radius), float(map_scale))[0].item())


_REF = None


def _reference_module():
    """The reference's own common/map_utils.py, loaded under a private name (never shadows this module)."""
    global _REF
    if _REF is not None:
        return _REF
    roots = [os.environ.get("DITREE_REFERENCE_ROOT"), os.getcwd(), *sys.path]
    for r in roots:
        if not r:
            continue
        cand = os.path.join(r, "common", "map_utils.py")
        if os.path.isfile(cand) and os.path.realpath(cand) != os.path.realpath(__file__):
            spec = importlib.util.spec_from_file_location("_ditree_reference_map_utils", cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _REF = mod
            return mod
    return None


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    ref = _reference_module()
    if ref is not None and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"common.map_utils.{name}: not provided by the engine and no reference checkout on the path")
