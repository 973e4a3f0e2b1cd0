"""End-to-end throughput of the drop-in RRT_Planner.plan() (host sampling + uploads + GPU rounds), candidates/s.
usage: python profiles/facade_throughput.py [batch ...]"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ditreeonlineplanner_amd.car_env import CarEnv                      # noqa: E402
from ditreeonlineplanner_amd.planners.RRT import RRT_Planner            # noqa: E402
from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler  # noqa: E402
from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net  # noqa: E402

maze = np.loadtxt(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ditreeonlineplanner_amd", "data",
                               "boxes.csv"), delimiter=",")
torch.manual_seed(0)
net = init_noise_pred_net(input_dim=2, action_dim=2, obs_dim=3, obs_history=1, action_history=1, goal_conditioned=True,
                          goal_dim=2, local_map_conditioned=True, local_map_encoder="resnet", local_map_embedding_dim=400,
                          local_map_size=20, down_dims=[512, 1024, 2048])
smp = DiffusionSampler(net, None, "carmaze", policy="flow_matching", pred_horizon=64, action_dim=2, prediction_type="actions",
                       obs_history=1, action_history=1, goal_conditioned=True, num_diffusion_iters=1, local_map_size=20).eval()
for batch in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    rounds = 12 if batch <= 2048 else 4
    env = CarEnv(maze_map=maze.copy(), collision_checking=False)
    start = np.array([*env.cell_rowcol_to_xy(np.array([17, 2])), np.deg2rad(45.0), 0.0, 0.0, 0.0])
    goal = np.array([*env.cell_rowcol_to_xy(np.array([2, 17])), 0, 0, 0, 0.0])
    n = batch * rounds
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=smp, prediction_type="actions", action_horizon=8,
                     local_map_size=20, local_map_scale=0.2, global_map_scale=1.0, goal_conditioning_bias=0.85,
                     prop_duration=[32], time_budget=600, batch=batch, max_candidates=n, capacity=1 << 17)
    for rep in range(2):                       # first pass warms up
        random.seed(1)
        np.random.seed(1)
        torch.manual_seed(1)
        pl.reset(start_state=start, goal_state=goal)
        pl._engine.env_goal = np.array([1e6, 1e6])          # unreachable goal: every round runs
        t0 = time.perf_counter()
        pl.plan()
        dt = time.perf_counter() - t0
    # the host side of a round: drawing its samples in the reference's RNG order (bulk: planners/_draw.py; the reference-order
    # loop for comparison) -- the draw of round r + 1 runs on a helper thread while round r is on the GPU
    random.seed(1)
    np.random.seed(1)
    t0 = time.perf_counter()
    for _ in range(5):
        pl.draw_round(batch, None)
    t_bulk = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    pl._draw_round_loop(batch, None)
    t_loop = time.perf_counter() - t0
    per_round = dt / rounds
    print(f"batch {batch}: {n} candidates in {dt * 1e3:.1f} ms = {n / dt:.0f} candidates/s (plan(), H=32, early exit on), "
          f"{pl.results['number_of_nodes']} nodes; round {per_round * 1e3:.1f} ms, host draw {t_bulk * 1e3:.2f} ms = "
          f"{100 * t_bulk / per_round:.1f} % of a round (reference-order loop: {t_loop * 1e3:.1f} ms = {100 * t_loop / per_round:.0f} %)")
