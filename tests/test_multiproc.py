"""N > 1 path.  CPU (gloo, world_size 2): the fixed-stride all-gather of candidate records and the
candidate partition.  GPU (-m gpu, 2 ranks sharing cuda:0 over gloo): a sharded round builds the
same tree, bit for bit, as the single-rank round."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import REPO, load_maze


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, B):
    sys.path.insert(0, REPO)
    from ditreeonlineplanner_amd.engine import allgather_round_fields
    _init(rank, world, port)
    per = (B + world - 1) // world
    total = per * world
    ref_i = torch.arange(total, dtype=torch.int32) * 3 + 1
    ref_f = torch.arange(total * 6, dtype=torch.float64).reshape(total, 6) * 0.5
    fi = torch.full((total,), -7, dtype=torch.int32)
    ff = torch.full((total, 6), -7.0, dtype=torch.float64)
    fi[rank * per:(rank + 1) * per] = ref_i[rank * per:(rank + 1) * per]
    ff[rank * per:(rank + 1) * per] = ref_f[rank * per:(rank + 1) * per]
    allgather_round_fields([fi, ff], per, rank, world)
    assert torch.equal(fi, ref_i) and torch.equal(ff, ref_f)
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 13])
def test_allgather_fields_gloo_cpu(B):
    mp.spawn(_cpu_worker, args=(2, 29531 + B, B), nprocs=2, join=True)


def test_allgather_fields_gloo_cpu_eight_ranks():
    """The rank count of the target node (8): fixed slots per rank, a ragged round (8190 = 7 x 1024 + 1022)."""
    mp.spawn(_cpu_worker, args=(8, 29571, 8190), nprocs=8, join=True)


def test_shard_partition():
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    for world in (1, 2, 4, 8):
        for B in (1, 7, 1024, 8190):
            got = []
            for r in range(world):
                e = ExpansionEngine.__new__(ExpansionEngine)
                e.rank, e.world = r, world
                lo, hi, per = e.shard(B)
                got.extend(range(lo, hi))
                assert hi - lo <= per
            assert got == list(range(B))


def _gpu_worker(rank, world, port, out_path, B=96):
    sys.path.insert(0, REPO)
    _init(rank, world, port)
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    from ditreeonlineplanner_amd.ops import Context
    from oracle import geometry as G
    from oracle import rrt as ORRT
    from oracle.tapes import ActionTape
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    ctx = Context(0)
    eng = ExpansionEngine(ctx, maze, start, goal, batch=B, capacity=4096, rank=rank, world_size=world)
    rt, at = ORRT.RandomTape(42), ActionTape(7)
    done = 0
    for _ in range(5):
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        acts = np.stack([at.actions(np.arange(done, done + B), j) for j in range(eng.n_chunks)], axis=1)
        eng.expand_round(torch.as_tensor(s).cuda(), torch.as_tensor(c).cuda(), inject_actions=torch.as_tensor(acts).cuda())
        done += B
    snap = eng.tree_snapshot()
    # the path walk is a collective in sharded mode: edges come from the ranks that expanded them
    node = eng.goal_node if eng.goal_node is not None else eng.fallback_node()
    path, actions = eng.path_to(node)
    owner = eng.tree.edge_owner[: len(snap["parents"])].cpu().numpy()
    np.savez(out_path.format(rank=rank), parents=snap["parents"], states=snap["states"], counters=snap["counters"],
             path=path, actions=actions, owner=owner, last_action=eng.tree.last_action[: len(snap["parents"])].cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_round_builds_identical_tree(tmp_path):
    out = str(tmp_path / "w{w}_r{rank}.npz")
    mp.spawn(_gpu_worker, args=(1, 29611, out.replace("{w}", "1")), nprocs=1, join=True)
    mp.spawn(_gpu_worker, args=(2, 29612, out.replace("{w}", "2")), nprocs=2, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    for r in range(2):
        b = np.load(out.replace("{w}", "2").format(rank=r))
        assert np.array_equal(a["parents"], b["parents"])
        assert np.array_equal(a["states"], b["states"])          # same kernels, same inputs: bit-identical
        assert np.array_equal(a["counters"][:5], b["counters"][:5])
        assert np.array_equal(a["last_action"], b["last_action"])    # carried by the 96-byte records
        assert np.array_equal(a["path"], b["path"]) and np.array_equal(a["actions"], b["actions"])
        assert set(np.unique(b["owner"][1:]).tolist()) <= {0, 1} and len(np.unique(b["owner"][1:])) == 2
    assert len(a["parents"]) > 50 and len(a["path"]) > 10


@pytest.mark.gpu
def test_sharded_round_four_ranks_ragged_batch(tmp_path):
    """Four ranks (the most one GPU box may host) and a round size that does not divide by the rank count (97 = 25 + 25 + 25
    + 22): fixed slots per rank, the last rank's tail slots unused -- the same tree, path and owners as one rank."""
    out = str(tmp_path / "q{w}_r{rank}.npz")
    mp.spawn(_gpu_worker, args=(1, 29651, out.replace("{w}", "1"), 97), nprocs=1, join=True)
    mp.spawn(_gpu_worker, args=(4, 29652, out.replace("{w}", "4"), 97), nprocs=4, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    for r in range(4):
        b = np.load(out.replace("{w}", "4").format(rank=r))
        assert np.array_equal(a["parents"], b["parents"]) and np.array_equal(a["states"], b["states"])
        assert np.array_equal(a["counters"][:5], b["counters"][:5]) and np.array_equal(a["last_action"], b["last_action"])
        assert np.array_equal(a["path"], b["path"]) and np.array_equal(a["actions"], b["actions"])
        assert set(np.unique(b["owner"][1:]).tolist()) <= {0, 1, 2, 3} and len(np.unique(b["owner"][1:])) >= 3
    assert len(a["parents"]) > 50


def _gpu_denoiser_worker(rank, world, port, out_path):
    """Sharded rounds with the real denoiser: every rank runs its sub-batch of candidates through the network."""
    sys.path.insert(0, REPO)
    _init(rank, world, port)
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context
    from oracle import geometry as G
    from oracle import rrt as ORRT
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    ctx = Context(0)
    B = 48                                   # 24 per rank: ragged against the 16-row padding of the denoiser batch
    net = NoisePredNet(seed=0)
    net.bind(ctx, precision=_lib.PREC_F16X3, max_batch=B)
    eng = ExpansionEngine(ctx, maze, start, goal, edge_length=32, batch=B, capacity=2048, rank=rank, world_size=world)
    rt = ORRT.RandomTape(42)
    g = torch.Generator().manual_seed(5)
    for _ in range(3):
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        noise = torch.randn(B, eng.n_chunks, 64, 2, generator=g)
        eng.expand_round(torch.as_tensor(s).cuda(), torch.as_tensor(c).cuda(), noise=noise.cuda())
    snap = eng.tree_snapshot()
    np.savez(out_path.format(rank=rank), parents=snap["parents"], states=snap["states"], counters=snap["counters"])
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_round_with_denoiser_builds_identical_tree(tmp_path):
    out = str(tmp_path / "d{w}_r{rank}.npz")
    mp.spawn(_gpu_denoiser_worker, args=(1, 29621, out.replace("{w}", "1")), nprocs=1, join=True)
    mp.spawn(_gpu_denoiser_worker, args=(2, 29622, out.replace("{w}", "2")), nprocs=2, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    for r in range(2):
        b = np.load(out.replace("{w}", "2").format(rank=r))
        assert np.array_equal(a["parents"], b["parents"])
        # rows of a denoiser batch are independent of the batch they run in (tests/test_gpu_fullsize.py): bit-identical
        assert np.array_equal(a["states"], b["states"])
        assert np.array_equal(a["counters"][:5], b["counters"][:5])


@pytest.mark.gpu
def test_native_rccl_allgather_single_rank():
    """The C-ABI communicator (ditree_comm_* / ditree_allgather_nodes) on one GPU: world 1 is the identity, but it opens
    librccl, builds a communicator and runs the collective on the stream (what a host without torch.distributed binds)."""
    import ctypes as C
    sys.path.insert(0, REPO)
    from ditreeonlineplanner_amd._lib import check, lib
    from ditreeonlineplanner_amd.ops import Context
    ctx = Context(0)
    uid = (C.c_uint8 * 128)()
    check(ctx._h, lib().ditree_comm_unique_id(ctx._h, uid), "comm_unique_id")
    check(ctx._h, lib().ditree_comm_init(ctx._h, 0, 1, uid), "comm_init")
    send = torch.arange(24, dtype=torch.float64, device="cuda") * 0.5
    recv = torch.zeros(24, dtype=torch.float64, device="cuda")
    check(ctx._h, lib().ditree_allgather_nodes(ctx._h, send.data_ptr(), recv.data_ptr(), 24, ctx.stream), "allgather_nodes")
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    assert lib().ditree_comm_init(ctx._h, 0, 1, uid) != 0        # a second communicator on the same ctx is refused
    check(ctx._h, lib().ditree_comm_destroy(ctx._h), "comm_destroy")
    ctx.close()


def _gpu_planner_worker(rank, world, port, out_path):
    """The drop-in surface sharded: RRT_Planner picks rank / world up from the initialised process group (what
    `torch.distributed.run -m ditreeonlineplanner_amd.run script.py` sets up); every rank seeds like the reference's scripts."""
    import random
    sys.path.insert(0, REPO)
    _init(rank, world, port)
    from ditreeonlineplanner_amd.engine import default_shard
    assert default_shard() == (0, 1, None)          # an initialised group alone is NOT adopted (opt-in, engine.default_shard)
    os.environ["DITREE_SHARD_DEFAULT_GROUP"] = "1"   # what `python -m ditreeonlineplanner_amd.run` sets when it joins the group
    from ditreeonlineplanner_amd.car_env import CarEnv
    from ditreeonlineplanner_amd.planners.RRT import RRT_Planner
    from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler
    from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net
    from ditreeonlineplanner_amd.common import map_utils
    torch.manual_seed(0)
    net = init_noise_pred_net(input_dim=2, action_dim=2, obs_dim=3, obs_history=1, action_history=1, goal_conditioned=True,
                              goal_dim=2, local_map_conditioned=True, local_map_encoder="resnet", local_map_embedding_dim=400,
                              local_map_size=20, down_dims=[512, 1024, 2048])
    with torch.no_grad():                     # damp the action head: more edges survive, the tree gets deep enough for a path
        for k, v in net.state_dict().items():
            if k.startswith("unet.final_conv.1"):
                v.mul_(0.05)
    smp = DiffusionSampler(net, None, "carmaze", policy="flow_matching", pred_horizon=64, action_dim=2, prediction_type="actions",
                           obs_history=1, action_history=1, goal_conditioned=True, num_diffusion_iters=1, local_map_size=20).eval()
    maze = load_maze("boxes")
    env = CarEnv(maze_map=maze, collision_checking=False)
    start = np.array([*env.cell_rowcol_to_xy(np.array([17, 2])), np.deg2rad(45.0), 0.0, 0.0, 0.0])
    goal = np.array([*env.cell_rowcol_to_xy(np.array([2, 17])), 0, 0, 0, 0.0])
    random.seed(42); np.random.seed(42); torch.manual_seed(42)                      # run_scenarios.py:86-90
    pl = RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=smp, action_horizon=8, local_map_size=20,
                     local_map_scale=0.2, global_map_scale=1.0, goal_conditioning_bias=0.85, prop_duration=[32],
                     time_budget=600, batch=48, max_candidates=144)
    assert (pl.rank, pl.world_size) == (rank, world)
    map_utils.cc_calls = 0
    pl.reset()
    path, actions = pl.plan()
    snap = pl._engine.tree_snapshot()
    np.savez(out_path.format(rank=rank), parents=snap["parents"], states=snap["states"], path=path, actions=actions,
             cc=map_utils.cc_calls, iters=pl.results["iterations"])
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_rrt_planner_returns_the_single_rank_path(tmp_path):
    out = str(tmp_path / "p{w}_r{rank}.npz")
    mp.spawn(_gpu_planner_worker, args=(1, 29631, out.replace("{w}", "1")), nprocs=1, join=True)
    mp.spawn(_gpu_planner_worker, args=(2, 29632, out.replace("{w}", "2")), nprocs=2, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    assert len(a["parents"]) > 8 and a["path"].shape[0] > 8
    for r in range(2):
        b = np.load(out.replace("{w}", "2").format(rank=r))
        assert np.array_equal(a["parents"], b["parents"]) and np.array_equal(a["states"], b["states"])
        assert np.array_equal(a["path"], b["path"]) and np.array_equal(a["actions"], b["actions"])      # bit for bit
        assert int(a["cc"]) == int(b["cc"]) and int(a["iters"]) == int(b["iters"])


def _gpu_mppi_worker(rank, world, port, out_path):
    """The MPPI controller sharded over ranks: every rank rolls out its block of the K rollouts (global rollout indices in
    the noise hash), two tiny all-reduces per step (MIN of beta, SUM of 3 + 2T doubles), replicated update and env step."""
    sys.path.insert(0, REPO)
    _init(rank, world, port)
    os.environ["DITREE_SHARD_DEFAULT_GROUP"] = "1"
    from ditreeonlineplanner_amd.mppi import MPPI
    from tests.test_gpu_mppi import l_path
    maze = load_maze("boxes")
    path, goal_xy = l_path(maze)
    m = MPPI(maze_data=maze, T=16, K=2048, nx=6, nu=2, seed=7)
    assert (m.rank, m.world) == (rank, world) and m.K_local == 2048 // world
    state = np.array([path[0, 0], path[0, 1], 0.0, 0.0, 0.0, 0.0])
    m.reset(start_state=state, goal_state=np.array([goal_xy[0], goal_xy[1], 0, 0, 0, 0.0]))
    m.set_ref_path(path)
    traj, acts = [], []
    for _ in range(60):
        state, a, done = m.step(state)
        assert done is False
        traj.append(state.copy()); acts.append(a.copy())
    np.savez(out_path.format(rank=rank), traj=np.array(traj), acts=np.array(acts), U=m._U.cpu().numpy(), ess=m.last["effective_samples"],
             coll=m.last["collided_rollouts"])
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_mppi_controller_follows_the_single_rank_trajectory(tmp_path):
    out = str(tmp_path / "m{w}_r{rank}.npz")
    mp.spawn(_gpu_mppi_worker, args=(1, 29641, out.replace("{w}", "1")), nprocs=1, join=True)
    mp.spawn(_gpu_mppi_worker, args=(2, 29642, out.replace("{w}", "2")), nprocs=2, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    b0, b1 = [np.load(out.replace("{w}", "2").format(rank=r)) for r in range(2)]
    # the ranks of a sharded run agree bit for bit (same all-reduced numbers, replicated update)
    assert np.array_equal(b0["traj"], b1["traj"]) and np.array_equal(b0["U"], b1["U"])
    # and follow the single-rank controller up to the order of the weighted sums (two partial sums added vs one pass)
    assert np.abs(a["traj"] - b0["traj"]).max() < 1e-9 and np.abs(a["acts"] - b0["acts"]).max() < 1e-9
    assert int(a["coll"]) == int(b0["coll"]) and abs(float(a["ess"]) - float(b0["ess"])) < 1e-6 * float(a["ess"])
    assert a["traj"][-1, 0] > a["traj"][0, 0] + 0.2                     # it moves along the corridor


def _gpu_ant_worker(rank, world, port, out_path, B=41):
    """Sharded ANT rounds (29-d tree, 134-double records incl. every candidate's end-of-edge history): tape dynamics + action tape."""
    sys.path.insert(0, REPO)
    _init(rank, world, port)
    from ditreeonlineplanner_amd.engine import AntExpansionEngine
    from ditreeonlineplanner_amd.ops import Context
    from oracle import ant as OA
    from oracle import rrt as ORRT
    from oracle import sampler as OS
    from tests.test_oracle_ant import trace_setup
    g, pre, pl, atape, otape, m = trace_setup("tape_boxes")
    md = OS.ANT_META
    norm = np.concatenate([md["Observations_mean"], md["Observations_std"], md["Actions_mean"], md["Actions_std"]])
    ctx = Context(0)
    eng = AntExpansionEngine(ctx, m["maze"], g[pre + "start"], g[pre + "goal"], desired_goal=g[pre + "desired"], norm=norm, batch=B,
                             capacity=2048, dynamics="tape", rank=rank, world_size=world, early_exit=True)
    assert eng.tree.record_doubles == 29 + 16 + 2 + 87
    tape = ORRT.RandomTape(42)
    done, goal = 0, None
    for _ in range(3):
        s, c = np.zeros((B, 29)), np.zeros((B, 2))
        for i in range(B):
            s[i], c[i] = OA.draw_candidate_ant(tape, 20, 20, 4.0, g[pre + "goal"])
        cand = np.arange(done, done + B)
        acts = np.stack([atape.actions(cand, j) for j in range(eng.n_chunks)], axis=1)
        cnt = eng.expand_round(torch.as_tensor(s).cuda(), torch.as_tensor(c).cuda(), inject_actions=torch.as_tensor(acts).cuda(),
                               next_obs_tape=torch.as_tensor(otape.rows(cand)).cuda())
        done += B
        if int(cnt[1]) >= 0:
            goal = int(cnt[1])
            break
    snap = eng.tree_snapshot()
    n = len(snap["parents"])
    node = goal if goal is not None else eng.fallback_node()
    path, actions = eng.path_to(node)
    np.savez(out_path.format(rank=rank), parents=snap["parents"], states=snap["states"], counters=snap["counters"], path=path,
             actions=actions, hist=eng.tree.hist[:n].cpu().numpy(), hist_n=eng.tree.hist_n[:n].cpu().numpy(),
             last_action=eng.tree.last_action[:n].cpu().numpy(), owner=eng.tree.edge_owner[:n].cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_ant_round_builds_identical_tree(tmp_path):
    """2 and 3 ranks (a ragged split of 41 candidates) build the 1-rank ant tree bit for bit -- node states, parents, the
    histories and last actions a child's first sampler call needs (carried by the records), and the collective path walk."""
    out = str(tmp_path / "a{w}_r{rank}.npz")
    mp.spawn(_gpu_ant_worker, args=(1, 29691, out.replace("{w}", "1")), nprocs=1, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    for world, port in ((2, 29692), (3, 29693)):
        mp.spawn(_gpu_ant_worker, args=(world, port, out.replace("{w}", str(world))), nprocs=world, join=True)
        for r in range(world):
            b = np.load(out.replace("{w}", str(world)).format(rank=r))
            for k in ("parents", "states", "hist", "hist_n", "last_action", "path", "actions"):
                assert np.array_equal(a[k], b[k]), (world, r, k)
            assert np.array_equal(a["counters"][:5], b["counters"][:5])
            assert len(np.unique(b["owner"][1:])) == world
    assert len(a["parents"]) > 10 and a["path"].shape[1] == 29
