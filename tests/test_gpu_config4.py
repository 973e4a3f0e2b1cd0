"""BASELINE config 4 as ONE tested workload (run_scenarios_with_lidar_DiTree.py:112-127 inside the expansion loop): a round of
8192 candidates with the denoiser in the loop (f16x3: local map -> conditioning -> encoder + U-Net -> 8 bicycle steps, x 4),
accept, then one 181-ray lidar scan of the TRUE maze per candidate end pose.

Full size: properties that do not need the oracle to run 8192 denoiser calls -- every recorded trajectory obeys the collision
/ goal rules it reports (checked state by state with the oracle's geometry), the tree is the accepted candidates in candidate
order, lidar end points lie in occupied cells of the true maze and visited cells are free.  A 64-candidate subset is compared
with the oracle round (fp32 torch-CPU denoiser) and with `oracle.geometry.lidar_scan`: candidates are independent, so the
subset's rows of the full round must equal the subset run through the oracle."""
import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import geometry as G
from oracle import rrt as ORRT
from oracle import sampler as OS
from tests.util import load_maze

pytestmark = pytest.mark.gpu
B, N0, H, A, P = 8192, 1024, 32, 8, 64


def test_config4_round_with_denoiser_in_the_loop_then_lidar():
    import bench
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.engine import CNT_GOAL, CNT_LATCH, CNT_NODES, ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context
    ctx = Context(0)
    dev = ctx.device
    maze = load_maze("boxes")
    true_maze = maze.copy()
    true_maze[8, 8:11] = 1                                      # an obstacle the planner does not know (the lidar's job)
    nodes, goal, samples, cond, noise = bench.synth_inputs(maze, B, seed=20260404)
    torch.manual_seed(0)
    onet = OD.init_noise_pred_net().eval()
    net = NoisePredNet(init=False)
    net.load_state_dict(onet.state_dict())
    net.bind(ctx, precision=_lib.PREC_F16X3, max_batch=B)
    eng = ExpansionEngine(ctx, maze, nodes[0], goal, edge_length=H, action_horizon=A, pred_horizon=P, batch=B,
                          capacity=N0 + B, emulate_sticky_done=False)
    t = eng.tree
    nd = torch.as_tensor(nodes, device=dev)
    t.state[:N0] = nd
    t.xy[:N0] = nd[:, :2]
    t.parent[:N0] = torch.arange(-1, N0 - 1, device=dev, dtype=torch.int32).clamp(min=0)
    t.parent[0] = -1
    t.has_prev[1:N0] = 1
    t.counters[CNT_NODES] = N0
    t.counters[CNT_GOAL] = -1
    t.counters[CNT_LATCH] = 0
    t.n_nodes_host = N0
    eng.expand_round(torch.as_tensor(samples, device=dev), torch.as_tensor(cond, device=dev), noise=noise.to(dev))
    rb = eng.rb
    status = rb.status[:B].cpu().numpy() & 0xFF
    parent = rb.parent[:B].cpu().numpy()
    chunks = rb.chunks_run[:B].cpu().numpy()
    steps = rb.chunk_steps[:B].cpu().numpy()
    states = rb.states[:B].cpu().numpy()
    end = rb.end_state[:B].cpu().numpy()
    node_id = rb.node_id[:B].cpu().numpy()

    # ---------------- full size: the recorded round obeys its own rules (geometry checked with the oracle's functions)
    assert set(np.unique(status)) <= {0, 1, 2} and chunks.min() >= 1 and chunks.max() <= H // A
    assert np.array_equal(parent, G.nn_argmin(samples[:, :2], nodes[:, :2]))
    assert (status == 2).sum() > B // 20 and (status == 0).sum() > B // 20
    last = chunks - 1
    k_last = steps[np.arange(B), last]
    assert ((steps > 0).sum(axis=1) == chunks).all()
    full = np.arange(H // A)[None, :] < last[:, None]                     # chunks before the last one ran all A steps, no event
    assert (steps[full] == A).all()
    assert (k_last[status == 0] == A).all() and (chunks[status == 0] == H // A).all()
    traj_end = states[np.arange(B), last, k_last]
    assert np.array_equal(traj_end, end)
    # the state a candidate ended on collides / is inside the goal radius exactly when its status says so ...
    coll_end = G.is_colliding_car(end, maze)
    goal_end = G.goal_reached(end, eng.env_goal)
    assert np.array_equal(coll_end, status == 2)
    assert np.array_equal(goal_end & ~coll_end, status == 1)
    # ... and no earlier state of any trajectory does (rows after the last step are zero rows: excluded)
    for j in range(H // A):
        for i in range(1, A + 1):
            ran = (j < last) | ((j == last) & (i < k_last))
            if not ran.any():
                continue
            s_ji = states[ran, j, i]
            assert not G.is_colliding_car(s_ji, maze).any() and not G.goal_reached(s_ji, eng.env_goal).any(), (j, i)
    # chunk boundaries are continuous
    for j in range(1, H // A):
        cont = j <= last
        assert np.array_equal(states[cont, j, 0], states[cont, j - 1, A])
    # accept: the tree is the non-collided candidates in candidate order, up to the first goal
    snap = eng.tree_snapshot()
    ok = np.nonzero(status != 2)[0]
    goal_rows = np.nonzero(status[ok] == 1)[0]
    if goal_rows.size:
        ok = ok[: goal_rows[0] + 1]
    assert len(snap["parents"]) == N0 + len(ok)
    assert np.array_equal(snap["parents"][N0:], parent[ok]) and np.array_equal(snap["states"][N0:], end[ok])
    assert np.array_equal(node_id[ok], N0 + np.arange(len(ok)))
    assert (eng.goal_node is not None) == bool(goal_rows.size)

    # ---------------- 64-candidate subset against the oracle round (fp32 denoiser on the CPU)
    rng = np.random.default_rng(3)
    sub = np.sort(rng.choice(B, 64, replace=False))
    nz = noise.numpy()

    def sampler(cand_idx, chunk, state, prev_action, has_prev, cond_goal, local_map):
        cv = OS.car_cond_vector(state, prev_action, has_prev, cond_goal)
        x1 = OS.flow_sample(onet, nz[sub[cand_idx], chunk], OS.scale_local_map(local_map), cv, k_steps=1)
        return OS.unnormalize_actions(x1)

    pl = ORRT.OraclePlanner(maze, nodes[0], goal, sampler, edge_length=H, action_horizon=A, emulate_sticky_done=False)
    tr = pl.tree
    for i in range(1, N0):
        tr.states.append(nodes[i].copy()); tr.parents.append(max(i - 1, 0)); tr.last_action.append(np.zeros(2))
        tr.has_prev.append(True); tr.num_visit.append(0); tr.edge_states.append(None); tr.edge_actions.append(None)
    ref = pl.expand_round(samples[sub], cond[sub])
    assert np.array_equal(status[sub], ref["status"]) and np.array_equal(chunks[sub], ref["chunks_run"])
    assert np.array_equal(steps[sub], ref["chunk_steps"]) and np.array_equal(parent[sub], ref["parent"])
    assert np.abs(end[sub] - ref["end_state"]).max() < 1e-5                   # the north star's state tolerance

    # ---------------- the lidar scan of the true maze from every end pose (run_scenarios_with_lidar_DiTree.py:112-121)
    Hh, W = maze.shape
    es = rb.end_state[:B]
    poses = torch.stack([es[:, 0] + W / 2, Hh / 2 - es[:, 1], es[:, 2]], dim=1).contiguous()       # (x_col, y_row, yaw) in cells
    dist, ends, hit, vis = ctx.lidar_scan(poses, torch.as_tensor(true_maze.astype(np.float32), device=dev))
    D, E, Hit, V = dist.cpu().numpy(), ends.cpu().numpy(), hit.cpu().numpy().astype(bool), vis.cpu().numpy()
    pn = poses.cpu().numpy()
    inside = (pn[:, 0] > 0) & (pn[:, 0] < W) & (pn[:, 1] > 0) & (pn[:, 1] < Hh)
    assert inside.all()                                              # collided candidates stop within a step of a wall
    ang = np.deg2rad(pn[:, 2:3] + G.LIDAR_ANGLES_DEG[None, :])
    assert np.abs(E[..., 0] - (pn[:, 0:1] + D * np.cos(ang))).max() < 1e-9
    assert np.abs(E[..., 1] - (pn[:, 1:2] + D * np.sin(ang))).max() < 1e-9
    ex = np.clip(np.floor(E[..., 0]).astype(int), 0, W - 1)
    ey = np.clip(np.floor(E[..., 1]).astype(int), 0, Hh - 1)
    assert (true_maze[ey[Hit], ex[Hit]] == 1).all()                  # a hit ends in an occupied cell of the TRUE maze
    assert (V.astype(bool) & (true_maze[None] == 1)).sum() == 0      # visited cells are free
    # scan_and_update_maze over all poses: the unknown obstacle is discovered by the candidates that see it
    known = maze.copy()
    known[ey[Hit], ex[Hit]] = 1
    assert (known[8, 8:11] == 1).all() and (known <= true_maze).all()
    for b in sub:
        d, e, v, h = G.lidar_scan(pn[b], true_maze)
        assert np.array_equal(Hit[b], h) and np.abs(D[b] - d).max() < 1e-9
        ref_v = np.zeros(maze.shape, dtype=np.uint8)
        ref_v[v[:, 1], v[:, 0]] = 1
        assert np.array_equal(V[b], ref_v)
    ctx.close()
