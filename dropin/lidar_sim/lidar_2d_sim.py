from ditreeonlineplanner_amd.lidar_sim.lidar_2d_sim import Lidar2DSim  # noqa: F401
