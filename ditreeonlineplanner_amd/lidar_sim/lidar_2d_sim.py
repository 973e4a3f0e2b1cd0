"""Lidar2DSim facade (reference: lidar_sim/lidar_2d_sim.py:5-98) on the HIP ray-march kernel.

``scan(robot_state, maze_data)`` keeps the reference's conventions (pose = (x_col, y_row, yaw)
in cell units, 181 rays at arange(-180, 182, 2) degrees, yaw in radians added to degrees).
Returned ``visited_points`` are the *set* of cells (x, y) a ray sample touched before its hit
(the reference returns the same cells as a ragged, duplicated list in ray order; the drivers only
use them as an index set, run_scenarios_with_lidar_DiTree.py:121).  ``scan_batch`` scans many poses
in one launch.  Gaussian range noise (noise_std > 0) is added on the host like lidar_2d_sim.py:30-33.
"""
from __future__ import annotations

import numpy as np
import torch


class Lidar2DSim:
    def __init__(self, azimuth_fov_deg=360, azimuth_res_deg=2.0, max_range=300, noise_std=0.0, scan_time=0.2,
                 ctx=None):
        if azimuth_fov_deg != 360 or azimuth_res_deg != 2.0 or max_range != 300:
            raise NotImplementedError("the HIP lidar kernel is built for the reference defaults (360 deg, 2 deg, 300)")
        self.azimuth_fov, self.azimuth_res = azimuth_fov_deg, azimuth_res_deg
        self.max_range, self.noise_std, self.scan_time = max_range, noise_std, scan_time
        self.angles_deg = np.arange(-self.azimuth_fov / 2, self.azimuth_fov / 2 + self.azimuth_res, self.azimuth_res)
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            from ..ops import default_context
            self._ctx = default_context()
        return self._ctx

    def scan_batch(self, poses, maze_data):
        """poses (B, 3) -> distances (B,181), endpoints (B,181,2), hit (B,181) bool, visited (B,R,C) bool."""
        ctx = self.ctx
        p = torch.as_tensor(np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3), device=ctx.device)
        m = torch.as_tensor(np.ascontiguousarray(maze_data, dtype=np.float32), device=ctx.device)
        d, e, h, v = ctx.lidar_scan(p, m, want_visited=True)
        return d.cpu().numpy(), e.cpu().numpy(), h.cpu().numpy().astype(bool), v.cpu().numpy().astype(bool)

    def scan(self, robot_state, maze_data, debug=False):
        pose = np.asarray(robot_state, dtype=np.float64)[:3]
        d, e, h, v = self.scan_batch(pose[None], maze_data)
        dist, ends = d[0], e[0]
        if self.noise_std > 0:
            dist = np.clip(dist + np.random.normal(0, self.noise_std, dist.shape), 0, self.max_range)
            ang = np.deg2rad(pose[2] + self.angles_deg)
            ends = np.stack([pose[0] + dist * np.cos(ang), pose[1] + dist * np.sin(ang)], axis=1)
        rc = np.argwhere(v[0])
        visited = rc[:, ::-1].copy()            # (x, y) = (col, row)
        return dist, ends, visited
