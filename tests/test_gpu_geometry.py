"""GPU parity: the HIP geometry kernels (through the C-ABI) against the CPU oracle and the
golden vectors captured from the reference.  Integer / flag / index outputs bit-exact;
FP64 states within 1e-9 abs (north-star bound 1e-5; device sin/cos/tanh differ from
glibc by ulps); f32 conditioning within 2e-6."""
import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import rrt as ORRT
from oracle import sampler as OS
from oracle.tapes import ActionTape
from tests.util import golden, load_maze

pytestmark = pytest.mark.gpu

FAR = np.array([1e6, 1e6])


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def test_collision_flags_bit_exact(ctx):
    g = golden("geometry")
    for name in ("Race_Track", "boxes", "random_huge", "narrow_short"):
        maze = load_maze(name)
        ctx.upload_maze(maze)
        poses = g[f"collision_{name}_poses"]
        st = np.zeros((len(poses), 6))
        st[:, :3] = poses
        state = dev(st)
        acts = torch.zeros(len(poses), 1, 2, dtype=torch.float64, device="cuda")
        status, states, _, steps = ctx.car_rollout(state, acts, FAR, A=1)
        got = (status.cpu().numpy() & 0xFF) == 2
        assert np.array_equal(got, g[f"collision_{name}_expected"]), name
        assert np.array_equal(states[:, 1].cpu().numpy(), st)        # v = 0, a = 0: state unchanged


def test_local_map_bit_exact(ctx):
    g = golden("geometry")
    for name in ("Race_Track", "boxes", "random_huge"):
        maze = load_maze(name).astype(np.float32)
        ctx.upload_maze(maze)
        for tag, (n, scale, sg) in {"car": (20, 0.2, 1.0), "ant": (16, 0.8, 4.0)}.items():
            poses = g[f"localmap_{name}_{tag}_poses"]
            st = np.zeros((len(poses), 6 if tag == "car" else 29))
            st[:, :3] = poses
            if tag == "ant":      # generated with positions scaled by s_global = 4 and theta = 0 (make_golden.gen_local_map)
                st[:, :2] *= sg
                st[:, 2] = 0.0
            out = ctx.local_map(dev(st), n=n, scale=scale, s_global=sg).cpu().numpy()
            exp = np.unpackbits(g[f"localmap_{name}_{tag}_expected"])[: out.size].reshape(out.shape)
            assert np.array_equal(out.astype(np.uint8), exp), (name, tag)
            out2 = ctx.local_map(dev(st), n=n, scale=scale, s_global=sg, scaled=True).cpu().numpy()
            assert np.array_equal(out2, exp.astype(np.float32) * 2 - 1)


def test_cond_vector(ctx):
    g = golden("network")
    out = ctx.cond_vector(dev(g["sampler_states"]), dev(g["sampler_prev"]),
                          dev(g["sampler_has_prev"].astype(np.uint8)), dev(g["sampler_goals"])).cpu().numpy()
    assert np.abs(out - g["sampler_cond_expected"]).max() < 2e-6
    assert np.array_equal(out[~g["sampler_has_prev"], 3:5], np.zeros((int((~g["sampler_has_prev"]).sum()), 2)))


def test_nn_argmin_bit_exact(ctx):
    g = golden("geometry")
    for n in (1, 17, 1000):
        idx = ctx.nn_argmin(dev(g[f"kdtree_{n}_queries"]), dev(g[f"kdtree_{n}_nodes"])).cpu().numpy()
        assert np.array_equal(idx, g[f"kdtree_{n}_expected"])
    # ties -> lowest index; ragged sizes
    rng = np.random.default_rng(3)
    nodes = rng.uniform(-5, 5, (777, 2))
    nodes[500] = nodes[20]
    nodes[300] = nodes[20]
    q = np.concatenate([nodes[20:21], rng.uniform(-5, 5, (130, 2))])
    idx = ctx.nn_argmin(dev(q), dev(nodes)).cpu().numpy()
    assert idx[0] == 20 and np.array_equal(idx, G.nn_argmin(q, nodes))


def test_lidar_matches_reference(ctx):
    g = golden("geometry")
    maze = load_maze("boxes")
    poses = g["lidar_boxes_poses"]
    dist, ends, hit, vis = ctx.lidar_scan(dev(poses), dev(maze, torch.float32))
    assert np.array_equal(hit.cpu().numpy().astype(bool), g["lidar_boxes_hit"])
    assert np.abs(dist.cpu().numpy() - g["lidar_boxes_dist"]).max() < 1e-9
    assert np.abs(ends.cpu().numpy() - g["lidar_boxes_end"]).max() < 1e-9
    exp_vis = np.unpackbits(g["lidar_boxes_visited"])[: len(poses) * maze.size].reshape(len(poses), *maze.shape)
    assert np.array_equal(vis.cpu().numpy(), exp_vis)
    # the reference's own demo scene (lidar_2d_sim.py:137-148)
    demo = np.zeros((100, 100), dtype=np.float32)
    demo[30:70, 40] = 1
    demo[50, 20:80] = 1
    demo[10:20, 10:20] = 1
    d, e, h, _ = ctx.lidar_scan(dev(np.array([[10.0, 80.0, 90.0]])), dev(demo))
    assert np.abs(d.cpu().numpy()[0] - g["lidar_demo_dist"]).max() < 1e-9
    assert np.abs(e.cpu().numpy()[0] - g["lidar_demo_end"]).max() < 1e-9


def test_dynamics_known_answers(ctx):
    g = golden("geometry")
    big = np.zeros((200, 200), dtype=np.float32)
    ctx.upload_maze(big)
    state = dev(g["dyn_s0"])
    acts = dev(g["dyn_actions"])
    status, states, aout, steps = ctx.car_rollout(state, acts, FAR, A=64)
    assert (status.cpu().numpy() == 0).all() and (steps.cpu().numpy() == 64).all()
    assert np.abs(states.cpu().numpy() - g["dyn_traj_expected"]).max() < 1e-9
    assert np.array_equal(aout.cpu().numpy(), g["dyn_actions"])


def test_dynamics_from_reference_source_text(ctx):
    """The rollout kernel against trajectories produced by the TEXT of the reference's CarEnv.step / _update_state /
    _check_done (golden dyntext_*): states to 1e-9 up to and including the step that reaches the goal, the goal flag at
    exactly that step (the reference freezes afterwards, the kernel stops and zero-fills)."""
    g = golden("geometry")
    ctx.upload_maze(np.zeros((200, 200), dtype=np.float32))
    S0, A, goals, traj, succ = (g[k] for k in ("dyntext_s0", "dyntext_actions", "dyntext_goals", "dyntext_traj_expected",
                                               "dyntext_success"))
    T = A.shape[1]
    skipped = 0
    for b in range(len(S0)):                      # one goal per launch
        if np.abs(traj[b, :, :2]).max() > 90.0:   # unbounded dynamics (no clamp on v): the run leaves any finite test map,
            skipped += 1                          # where the kernel reports a collision and the text (no map) goes on
            continue
        state = dev(S0[b:b + 1])
        status, states, aout, steps = ctx.car_rollout(state, dev(A[b:b + 1]), goals[b], A=T)
        st, n = int(status.item()) & 0xFF, int(steps.item())
        first = int(np.argmax(succ[b])) + 1 if succ[b].any() else None
        if first is None:
            assert st == 0 and n == T
        else:
            assert st == 1 and n == first, (b, st, n, first)
        got = states.cpu().numpy()[0]
        assert np.abs(got[: n + 1] - traj[b, : n + 1]).max() < 1e-9
        assert np.array_equal(got[n + 1:], np.zeros_like(got[n + 1:]))
    assert skipped <= 2


@pytest.mark.parametrize("name", ["boxes", "Race_Track", "random_huge"])
def test_rollout_chunk_vs_oracle(ctx, name):
    maze = load_maze(name)
    ctx.upload_maze(maze)
    rng = np.random.default_rng(11)
    B = 4096
    free = np.argwhere(maze == 0)
    cell = free[rng.integers(0, len(free), B)]
    xy = G.cell_rowcol_to_xy(cell, maze) + rng.uniform(-0.45, 0.45, (B, 2))
    st = np.concatenate([xy, rng.uniform(-np.pi, np.pi, (B, 1)), rng.uniform(0, 4, (B, 1)),
                         rng.uniform(0, 1, (B, 1)), rng.uniform(-0.4, 0.4, (B, 1))], axis=1)
    acts = np.stack([rng.uniform(-12, 12, (B, 8)), rng.uniform(-3, 3, (B, 8))], axis=2)
    goal = G.cell_rowcol_to_xy(free[len(free) // 2], maze)
    exp = G.rollout_chunk(st, acts, maze, goal, 8)
    state = dev(st)
    status, states, aout, steps = ctx.car_rollout(state, dev(acts), goal, A=8)
    s = status.cpu().numpy()
    assert np.array_equal(s & 0xFF, exp["status"])
    assert np.array_equal((s & 0x100) != 0, exp["goal_at_collision"])
    assert np.array_equal(steps.cpu().numpy(), exp["n_steps"])
    assert np.abs(states.cpu().numpy() - exp["states"]).max() < 1e-9
    assert np.abs(state.cpu().numpy() - exp["end_state"]).max() < 1e-9
    assert np.array_equal(aout.cpu().numpy(), exp["actions"])
    assert (exp["status"] == 2).sum() > 100 and (exp["status"] == 1).sum() > 0


def _scenario(g, tag):
    maze = load_maze(str(g[f"trace_{tag}_maze_name"]))
    sr, sc, sdeg, gr, gc = [int(v) for v in g[f"trace_{tag}_scenario"]]
    start = np.array([*G.cell_rowcol_to_xy([sr, sc], maze), np.deg2rad(float(sdeg)), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([gr, gc], maze), 0, 0, 0, 0])
    return maze, start, goal


def _run_engine(ctx, maze, start, goal, tape_seed, budget, batch, early_exit=False):
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    eng = ExpansionEngine(ctx, maze, start, goal, batch=batch, capacity=4096, early_exit=early_exit)
    rt = ORRT.RandomTape(42)
    at = ActionTape(tape_seed)
    done = 0
    while eng.goal_node is None and done < budget:
        B = min(batch, budget - done)
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        acts = np.stack([at.actions(np.arange(done, done + B), j) for j in range(eng.n_chunks)], axis=1)
        eng.expand_round(dev(s), dev(c), inject_actions=dev(acts))
        done += B
    return eng


@pytest.mark.parametrize("tag", ["race", "boxes", "rlarge2", "easy"])
def test_engine_b1_reproduces_reference_trace(ctx, tag):
    """B = 1 rounds with the golden tapes: tree parents / reached flag bit-exact vs the reference planner."""
    g = golden("traces")
    maze, start, goal = _scenario(g, tag)
    eng = _run_engine(ctx, maze, start, goal, int(g[f"trace_{tag}_tape_seed"]), int(g[f"trace_{tag}_budget"]), 1)
    snap = eng.tree_snapshot()
    assert np.array_equal(snap["parents"], g[f"trace_{tag}_parents"])
    assert np.abs(snap["states"] - g[f"trace_{tag}_states"]).max() < 1e-9
    reached = eng.goal_node is not None
    assert reached == bool(g[f"trace_{tag}_reached"])
    assert int(snap["counters"][3]) == int(g[f"trace_{tag}_iterations"])
    assert int(snap["counters"][5]) == 0
    node = eng.goal_node if reached else eng.fallback_node()
    path, actions = eng.path_to(node)
    assert path.shape == g[f"trace_{tag}_path"].shape and actions.shape == g[f"trace_{tag}_actions"].shape
    assert np.abs(path - g[f"trace_{tag}_path"]).max() < 1e-5
    assert np.array_equal(actions, g[f"trace_{tag}_actions"])


@pytest.mark.parametrize("batch", [7, 64, 256])
@pytest.mark.parametrize("early_exit", [False, True])
def test_engine_rounds_vs_oracle_rounds(ctx, batch, early_exit):
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    budget = batch * 6
    pl = ORRT.OraclePlanner(maze, start, goal, ActionTape(99).sampler())
    reached, path, actions = pl.plan(ORRT.RandomTape(42), budget, batch=batch)
    eng = _run_engine(ctx, maze, start, goal, 99, budget, batch, early_exit)
    snap = eng.tree_snapshot()
    assert np.array_equal(snap["parents"], np.array(pl.tree.parents))
    assert np.abs(snap["states"] - np.array(pl.tree.states)).max() < 1e-9
    assert (eng.goal_node is not None) == reached
    assert int(snap["counters"][3]) == pl.iterations and int(snap["counters"][4]) == pl.candidates
    nv = eng.tree.num_visit[: len(pl.tree)].cpu().numpy()
    assert np.array_equal(nv, np.array(pl.tree.num_visit))


def _run_engine_sched(ctx, maze, start, goal, tape_seed, budget, batch, sched, early_exit=False):
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    eng = ExpansionEngine(ctx, maze, start, goal, batch=batch, capacity=4096, early_exit=early_exit, prop_duration=list(sched))
    rt = ORRT.RandomTape(42)
    at = ActionTape(tape_seed)
    done = 0
    while eng.goal_node is None and done < budget:
        B = min(batch, budget - done)
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        acts = np.stack([at.actions(np.arange(done, done + B), j) for j in range(eng.n_chunks)], axis=1)
        eng.expand_round(dev(s), dev(c), inject_actions=dev(acts))
        done += B
    return eng


def test_prop_duration_schedule_b1_reproduces_reference_trace(ctx):
    """planners/RRT.py:149-152 with prop_duration = [128, 64, 32]: the edge length of a visit follows the parent's visit
    count.  B = 1 rounds = the reference planner (golden `sched_*` from its own run)."""
    g = golden("traces")
    maze = load_maze("boxes")
    sched = [int(v) for v in g["sched_schedule"]]
    eng = _run_engine_sched(ctx, maze, g["sched_start"], g["sched_goal"], int(g["sched_seed"]), int(g["sched_budget"]), 1, sched)
    snap = eng.tree_snapshot()
    assert np.array_equal(snap["parents"], g["sched_parents"])
    assert np.abs(snap["states"] - g["sched_states"]).max() < 1e-9
    assert (eng.goal_node is not None) == bool(g["sched_reached"])
    assert int(snap["counters"][3]) == int(g["sched_iterations"])
    na = eng.tree.edge_nactions[: len(g["sched_parents"])].cpu().numpy()
    assert np.array_equal(na[1:], g["sched_edge_actions"][1:]) and len(set(na[1:].tolist())) > 1
    node = eng.goal_node if eng.goal_node is not None else eng.fallback_node()
    path, actions = eng.path_to(node)
    assert np.abs(path - g["sched_path"]).max() < 1e-5 and np.array_equal(actions, g["sched_actions"])


@pytest.mark.parametrize("batch", [16, 128])
@pytest.mark.parametrize("early_exit", [False, True])
def test_prop_duration_schedule_rounds_vs_oracle(ctx, batch, early_exit):
    """Rounds of B > 1: the candidates of one parent are its visits in candidate order (oracle round semantics)."""
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    sched = [96, 64, 32, 16]
    budget = batch * 5
    pl = ORRT.OraclePlanner(maze, start, goal, ActionTape(31).sampler(), prop_duration=sched)
    reached, path, actions = pl.plan(ORRT.RandomTape(42), budget, batch=batch)
    eng = _run_engine_sched(ctx, maze, start, goal, 31, budget, batch, sched, early_exit)
    snap = eng.tree_snapshot()
    assert np.array_equal(snap["parents"], np.array(pl.tree.parents))
    assert np.abs(snap["states"] - np.array(pl.tree.states)).max() < 1e-9
    assert (eng.goal_node is not None) == reached
    assert int(snap["counters"][3]) == pl.iterations and int(snap["counters"][4]) == pl.candidates
    assert np.array_equal(eng.tree.num_visit[: len(pl.tree)].cpu().numpy(), np.array(pl.tree.num_visit))
    lens = np.array([0 if e is None else len(e) for e in pl.tree.edge_actions])
    assert np.array_equal(eng.tree.edge_nactions[: len(pl.tree)].cpu().numpy()[1:], lens[1:])
