"""MI355X-native DiTree expansion engine (HIP kernels behind a C-ABI, Python host).

Public surface mirrors the reference's modules for the hot path:
  ditreeonlineplanner_amd.planners.RRT.RRT_Planner
  ditreeonlineplanner_amd.policies.fm_policy.DiffusionSampler
  ditreeonlineplanner_amd.car_env.CarEnv
  ditreeonlineplanner_amd.lidar_sim.lidar_2d_sim.Lidar2DSim
  ditreeonlineplanner_amd.train_diffusion_policy.init_noise_pred_net
plus the engine itself (ditreeonlineplanner_amd.engine).
"""
__version__ = "0.1.0"
