"""GPU: edge cases of the C-ABI -- empty and ragged batches, argument / call-order errors,
tree capacity overflow, maze limits."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import rrt as ORRT
from oracle.tapes import ActionTape
from tests.util import load_maze

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def test_call_order_errors(ctx):
    from ditreeonlineplanner_amd._lib import DitreeError
    st = torch.zeros(4, 6, dtype=torch.float64, device="cuda")
    with pytest.raises(DitreeError, match="no maze"):
        ctx.local_map(st)
    with pytest.raises(DitreeError, match="no maze"):
        ctx.car_rollout(st, torch.zeros(4, 8, 2, dtype=torch.float64, device="cuda"), np.zeros(2))
    with pytest.raises(DitreeError, match="not loaded"):
        ctx.denoise(torch.zeros(4, 64, 2, device="cuda"), torch.zeros(4, 20, 20, device="cuda"), torch.zeros(4, 7, device="cuda"))
    with pytest.raises(DitreeError):
        ctx.upload_maze(np.zeros((300, 300), dtype=np.float32))          # beyond the LDS-staged limit
    with pytest.raises(TypeError):
        ctx.local_map(st.float())                                        # wrong dtype never reaches the kernels


def test_empty_batches_are_noops(ctx):
    maze = load_maze("boxes")
    ctx.upload_maze(maze)
    e6 = torch.zeros(0, 6, dtype=torch.float64, device="cuda")
    assert ctx.local_map(e6).shape == (0, 20, 20)
    status, states, aout, steps = ctx.car_rollout(e6, torch.zeros(0, 8, 2, dtype=torch.float64, device="cuda"), np.zeros(2))
    assert status.numel() == 0 and states.shape == (0, 9, 6)
    idx = ctx.nn_argmin(torch.zeros(0, 2, dtype=torch.float64, device="cuda"), torch.zeros(3, 2, dtype=torch.float64, device="cuda"))
    assert idx.numel() == 0
    d, e, h, v = ctx.lidar_scan(torch.zeros(0, 3, dtype=torch.float64, device="cuda"), dev(maze, torch.float32))
    assert d.shape == (0, 181)


@pytest.mark.parametrize("B", [1, 3, 63, 65, 257])
def test_ragged_batches_match_oracle(ctx, B):
    maze = load_maze("Race_Track")
    ctx.upload_maze(maze)
    rng = np.random.default_rng(B)
    free = np.argwhere(maze == 0)
    xy = G.cell_rowcol_to_xy(free[rng.integers(0, len(free), B)], maze) + rng.uniform(-0.4, 0.4, (B, 2))
    st = np.concatenate([xy, rng.uniform(-3, 3, (B, 1)), rng.uniform(0, 4, (B, 1)), rng.uniform(0, 1, (B, 1)),
                         rng.uniform(-0.4, 0.4, (B, 1))], axis=1)
    acts = np.stack([rng.uniform(-12, 12, (B, 8)), rng.uniform(-3, 3, (B, 8))], axis=2)
    goal = G.cell_rowcol_to_xy(np.array([1, 10]), maze)
    ref = G.rollout_chunk(st, acts, maze, goal, 8)
    state = dev(st)
    status, states, aout, steps = ctx.car_rollout(state, dev(acts), goal, A=8)
    assert np.array_equal(status.cpu().numpy() & 0xFF, ref["status"])
    assert np.abs(states.cpu().numpy() - ref["states"]).max() < 1e-9
    lm = ctx.local_map(dev(st)).cpu().numpy()
    assert np.array_equal(lm, G.create_local_map(maze.astype(np.float32), st[:, 0], st[:, 1], st[:, 2], 20, 0.2, 1.0,
                                                 (maze.shape[1] / 2, maze.shape[0] / 2)))
    nodes = rng.uniform(-3, 3, (B, 2))
    assert np.array_equal(ctx.nn_argmin(dev(st[:, :2].copy()), dev(nodes)).cpu().numpy(), G.nn_argmin(st[:, :2], nodes))


def test_capacity_overflow_is_flagged_not_faulting(ctx):
    from ditreeonlineplanner_amd.engine import CNT_NODES, CNT_OVERFLOW, ExpansionEngine
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    eng = ExpansionEngine(ctx, maze, start, goal, batch=64, capacity=24, emulate_sticky_done=False)
    from oracle import rrt as ORRT
    rt, at = ORRT.RandomTape(42), ActionTape(99)
    done = 0
    for _ in range(6):
        s, c = rt.draw_round(64, 20, 20, goal)
        acts = np.stack([at.actions(np.arange(done, done + 64), j) for j in range(eng.n_chunks)], axis=1)
        cnt = eng.expand_round(dev(s), dev(c), inject_actions=dev(acts))
        done += 64
    assert int(cnt[CNT_NODES]) == 24 and int(cnt[CNT_OVERFLOW]) == 1
    snap = eng.tree_snapshot()
    assert (snap["parents"][1:] >= 0).all() and (snap["parents"] < 24).all()


def test_denoise_argument_checks(ctx):
    from ditreeonlineplanner_amd._lib import DitreeError
    from ditreeonlineplanner_amd.model import NoisePredNet
    net = NoisePredNet(seed=1)
    net.bind(ctx, precision=0, max_batch=16)
    n = torch.zeros(32, 64, 2, device="cuda")
    with pytest.raises(DitreeError, match="exceeds"):
        ctx.denoise(n, torch.zeros(32, 20, 20, device="cuda"), torch.zeros(32, 7, device="cuda"))
    ctx.denoise_reserve(32, 0)                                            # growing the workspace is allowed
    out = ctx.denoise(n, torch.zeros(32, 20, 20, device="cuda"), torch.zeros(32, 7, device="cuda"), want_actions=False)
    assert torch.isfinite(out).all()


def test_round_rejects_mismatched_horizon_and_map(ctx):
    """ditree_expand_round sizes its scratch from the caller's pred_horizon / local-map size while the denoiser strides
    them by ITS dimensions: a mismatch is a clean DITREE_E_ARG, never an out-of-bounds write."""
    from ditreeonlineplanner_amd._lib import DitreeError
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), 0.7, 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    net = NoisePredNet(seed=1)
    net.bind(ctx, precision=0, max_batch=16)
    s = torch.zeros(8, 6, dtype=torch.float64, device="cuda")
    c = torch.zeros(8, 2, dtype=torch.float64, device="cuda")
    eng = ExpansionEngine(ctx, maze, start, goal, edge_length=32, pred_horizon=32, batch=8, capacity=64)
    with pytest.raises(DitreeError, match="pred_horizon 32"):
        eng.expand_round(s, c, noise=torch.zeros(8, 4, 32, 2, device="cuda"))
    eng = ExpansionEngine(ctx, maze, start, goal, edge_length=32, local_map_size=16, batch=8, capacity=64)
    with pytest.raises(DitreeError, match="local map 16"):
        eng.expand_round(s, c, noise=torch.zeros(8, 4, 64, 2, device="cuda"))
    # tensors of the wrong shape never reach the library
    with pytest.raises(ValueError):
        ctx.denoise(torch.zeros(8, 32, 2, device="cuda"), torch.zeros(8, 20, 20, device="cuda"), torch.zeros(8, 7, device="cuda"))
    with pytest.raises(ValueError):
        ctx.denoise(torch.zeros(8, 64, 2, device="cuda"), torch.zeros(8, 16, 16, device="cuda"), torch.zeros(8, 7, device="cuda"))


def test_weight_blob_checksum_is_verified(ctx):
    from ditreeonlineplanner_amd._lib import DitreeError
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.weights import pack_state_dict
    net = NoisePredNet(seed=2)
    blob, manifest = pack_state_dict(net.state_dict(), pred_horizon=64, local_map_size=20)
    assert "#checksum" in manifest and "#config pred_horizon 64 local_map_size 20" in manifest
    bad = blob.copy()
    bad[12345] += 1.0
    with pytest.raises(DitreeError, match="checksum"):
        ctx.load_weights(bad, manifest)
    ctx.load_weights(blob, manifest)


def test_shared_context_keeps_each_planners_maze_and_each_nets_weights(ctx):
    """Two engines with different mazes and two nets share one ctx (what the facades' default_context does): every launch
    re-uploads its owner's maze / weights when somebody else used the ctx in between."""
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    m1, m2 = load_maze("boxes"), load_maze("random_huge")

    def mk(maze, rc0, rc1):
        start = np.array([*G.cell_rowcol_to_xy(rc0, maze), 0.7, 0, 0, 0])
        goal = np.array([*G.cell_rowcol_to_xy(rc1, maze), 0, 0, 0, 0])
        return ExpansionEngine(ctx, maze, start, goal, edge_length=32, batch=64, capacity=1024), start, goal
    e1, s1, g1 = mk(m1, [17, 2], [2, 17])
    rt, at = ORRT.RandomTape(42), ActionTape(5)
    s, c = rt.draw_round(64, m1.shape[1], m1.shape[0], g1)
    acts = np.stack([at.actions(np.arange(64), j) for j in range(e1.n_chunks)], axis=1)
    args = (torch.as_tensor(s).cuda(), torch.as_tensor(c).cuda())
    e1.expand_round(*args, inject_actions=torch.as_tensor(acts).cuda())
    ref = e1.tree_snapshot()
    e1b, _, _ = mk(m1, [17, 2], [2, 17])
    e2, _, _ = mk(m2, [1, 1], [29, 29])                  # uploads ANOTHER maze into the shared ctx
    assert ctx.maze_owner is e2
    e1b.expand_round(*args, inject_actions=torch.as_tensor(acts).cuda())
    assert ctx.maze_owner is e1b
    again = e1b.tree_snapshot()
    assert np.array_equal(ref["parents"], again["parents"]) and np.array_equal(ref["states"], again["states"])
    # weights: the second bind replaces the first net's device copy; the first net notices and re-binds
    n1, n2 = NoisePredNet(seed=1), NoisePredNet(seed=2)
    x = torch.randn(4, 64, 2)
    lm = torch.zeros(4, 20, 20)
    cd = torch.zeros(4, 7)
    n1.bind(ctx, precision=0, max_batch=16)
    y1 = n1(x, lm, torch.zeros(4), cd).cpu()
    n2.bind(ctx, precision=0, max_batch=16)
    y2 = n2(x, lm, torch.zeros(4), cd).cpu()
    assert not torch.equal(y1, y2) and not n1.is_current(ctx)
    assert torch.equal(n1(x, lm, torch.zeros(4), cd).cpu(), y1)
    # in-place parameter updates after bind are picked up too
    with torch.no_grad():
        n1.get_parameter("unet.final_conv.1.bias").add_(0.5)        # (a scale on a conv in front of a GroupNorm would cancel)
    assert not n1.is_current(ctx)
    assert not torch.equal(n1(x, lm, torch.zeros(4), cd).cpu(), y1)


def test_round3_entry_points_argument_and_call_order_errors(ctx):
    """ditree_mppi_step / ditree_expand_round_ant / ditree_path_after_obstacle / ditree_denoise_status: bad arguments and
    call order give error codes with a message -- never a fault, never a silent no-op."""
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd._lib import DitreeError, check, lib
    from ditreeonlineplanner_amd.mppi import MPPI
    maze = load_maze("boxes")
    f64 = torch.float64
    # status before any weights: nothing can have saturated
    assert ctx.denoise_status() == []
    # path_after_obstacle before a maze
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    path = torch.zeros(8, 6, dtype=torch.float32, device="cuda")
    cur = (C.c_double * 2)(0.0, 0.0)
    with pytest.raises(DitreeError, match="no maze"):
        check(ctx._h, lib().ditree_path_after_obstacle(ctx._h, path.data_ptr(), 6, 8, cur, 1, out.data_ptr(), ctx.stream), "path_after_obstacle")
    ctx.upload_maze(maze)
    with pytest.raises(DitreeError, match="bad argument"):
        check(ctx._h, lib().ditree_path_after_obstacle(ctx._h, path.data_ptr(), 1, 8, cur, 1, out.data_ptr(), ctx.stream), "path_after_obstacle")
    # a path that crosses nothing: k = -1 (the reference keeps its last point); a path that ends inside an obstacle: k = len
    free_row = np.stack([np.linspace(-8.5, 8.5, 40), np.full(40, -8.5)], axis=1).astype(np.float32)           # bottom corridor
    p_dev = dev(free_row)
    check(ctx._h, lib().ditree_path_after_obstacle(ctx._h, p_dev.data_ptr(), 2, 40, (C.c_double * 2)(-8.5, -8.5), 0, out.data_ptr(), ctx.stream), "x")
    assert out.cpu().tolist() == [0, -1]
    into_wall = np.stack([np.linspace(-8.5, 9.5, 40), np.full(40, -8.5)], axis=1).astype(np.float32)         # ends in the border wall
    check(ctx._h, lib().ditree_path_after_obstacle(ctx._h, dev(into_wall).data_ptr(), 2, 40, (C.c_double * 2)(-8.5, -8.5), 0, out.data_ptr(), ctx.stream), "x")
    c, k = out.cpu().tolist()
    assert c == 0 and k == 40
    # MPPI: construction limits, step before a reference path, bad stage mask / lanes through the C-ABI
    with pytest.raises(ValueError):
        MPPI(maze_data=maze, T=65, K=8, ctx=ctx)
    with pytest.raises(NotImplementedError):
        MPPI(maze_data=maze, T=8, K=8, nx=10, nu=4, ctx=ctx)
    m = MPPI(maze_data=maze, T=8, K=16, ctx=ctx)
    with pytest.raises(DitreeError, match="set_ref_path"):
        m.step(np.zeros(6))
    m.set_ref_path(free_row.astype(np.float64))
    m.reset(start_state=np.array([-8.5, -8.5, 0, 0, 0, 0.0]), goal_state=np.array([8.5, -8.5, 0, 0, 0, 0.0]))
    for stages in (0, 32, 64):
        with pytest.raises(DitreeError, match="stages"):
            m.launch(stages)
    m.params.lanes = 3
    with pytest.raises(DitreeError, match="lanes"):
        m.launch(_lib.MPPI_ALL)
    m.params.lanes = 0
    m.params.lam = 0.0
    with pytest.raises(DitreeError, match="lambda"):
        m.launch(_lib.MPPI_ALL)
    m.params.lam = 1.0
    nxt, a, done = m.step(np.array([-8.5, -8.5, 0, 0, 0, 0.0]))
    assert done is False and np.isfinite(nxt).all()
    # the ant round without the ant network / with malformed inputs
    from ditreeonlineplanner_amd.engine import AntExpansionEngine
    z = lambda *shape, dt=f64: torch.zeros(*shape, dtype=dt, device="cuda")      # noqa: E731
    st = np.zeros(29)
    st[2], st[3] = 0.75, 1.0
    eng = AntExpansionEngine(ctx, maze, st, st, norm=np.ones(70), batch=4, capacity=64, edge_length=4, dynamics="tape")
    with pytest.raises(DitreeError, match="not loaded"):
        eng.expand_round(z(4, 29), z(4, 2), noise=z(4, 2, 16, 8, dt=torch.float32), next_obs_tape=z(4, 2, 2, 29))
    with pytest.raises(ValueError, match="next_obs_tape"):
        eng.expand_round(z(4, 29), z(4, 2), inject_actions=z(4, 2, 16, 8))
    with pytest.raises(ValueError, match="samples must be"):
        eng.expand_round(z(4, 6), z(4, 2), inject_actions=z(4, 2, 16, 8), next_obs_tape=z(4, 2, 2, 29))
    # a car tree handed to the ant round, and the reverse
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    car = ExpansionEngine(ctx, maze, np.zeros(6), np.zeros(6), batch=4, capacity=64)
    rp, keep = eng._params(z(4, 29), z(4, 2), None, z(4, 2, 16, 8), 0, 4, z(4, 2, 2, 29))
    with pytest.raises(DitreeError, match="state_dim 29"):
        check(ctx._h, lib().ditree_expand_round_ant(ctx._h, C.byref(car.tree.desc), C.byref(car.rb.desc(0, 4)), C.byref(rp), ctx.stream), "x")
    with pytest.raises(DitreeError, match="sticky-done"):
        check(ctx._h, lib().ditree_accept(ctx._h, C.byref(eng.tree.desc), C.byref(eng.rb.desc(0, 4)), 1, ctx.stream), "x")


def test_rounds_on_a_side_stream_equal_rounds_on_the_default_stream(ctx):
    """Every entry point enqueues on the stream it is handed (torch's CURRENT stream through the front end): an expansion on a
    non-default stream gives the same tree, and is ordered with respect to that stream only."""
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    trees = []
    for side in (False, True):
        eng = ExpansionEngine(ctx, maze, start, goal, batch=64, capacity=2048)
        rt, at = ORRT.RandomTape(42), ActionTape(5)
        stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            done = 0
            for _ in range(4):
                s, c = rt.draw_round(64, maze.shape[1], maze.shape[0], goal)
                acts = np.stack([at.actions(np.arange(done, done + 64), j) for j in range(eng.n_chunks)], axis=1)
                eng.expand_round(dev(s), dev(c), inject_actions=dev(acts))
                done += 64
            stream.synchronize()
            trees.append(eng.tree_snapshot())
    assert np.array_equal(trees[0]["parents"], trees[1]["parents"]) and np.array_equal(trees[0]["states"], trees[1]["states"])
    assert len(trees[0]["parents"]) > 20


def test_round3_kernels_at_their_smallest_and_odd_sizes(ctx):
    """Degenerate sizes of the new kernels: one rollout / one step / a one-point path, odd rollout counts across the lane
    groups and slices, a one-candidate ant round, a one-point reference path -- results finite and consistent, no fault."""
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd._lib import check, lib
    from ditreeonlineplanner_amd.mppi import MPPI
    from oracle import mppi as OM
    maze = load_maze("boxes")
    start = np.array([-8.5, -8.5, 0.0, 1.0, 0.2, 0.0])
    goal = np.array([8.5, -8.5, 0, 0, 0, 0.0])
    for K, T, P, lanes in ((1, 1, 1, 1), (3, 5, 2, 2), (127, 7, 33, 4), (257, 16, 100, 2), (1000, 64, 4096, 4)):
        m = MPPI(maze_data=maze, T=T, K=K, lanes=lanes, seed=K, ctx=ctx)
        m.reset(start_state=start, goal_state=goal)
        path = np.stack([np.linspace(-8.5, 8.5, P), np.full(P, -8.5)], axis=1)
        m.set_ref_path(path)
        nxt, a, done = m.step(start)
        assert done is False and np.isfinite(nxt).all() and np.isfinite(a).all(), (K, T, P)
        assert 0 < m.last["effective_samples"] <= K + 1e-9 and m.last["eta"] >= 1.0 - 1e-12
        # the same call on the numpy restatement (device noise mirrored on the host)
        kw = dict(lam=1.0, sigma=(3.0, 0.6), w_track=20.0, w_progress=0.5, w_collision=1e3, w_goal=50.0, window_back=8, window_fwd=56)
        eps = OM.device_noise(K, 0, K, T, kw["sigma"])
        rc, rf, _ = OM.rollout_costs(maze, start, np.zeros((T, 2)), path, m.env.goal, eps, **kw)
        assert np.abs(m._costs.cpu().numpy() - rc).max() < 1e-9 * max(1.0, np.abs(rc).max()), (K, T, P)
    # one-point reference path for extract_path_after_obstacle
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    ctx.upload_maze(maze)
    one = dev(np.array([[-8.5, -8.5]], dtype=np.float32))
    check(ctx._h, lib().ditree_path_after_obstacle(ctx._h, one.data_ptr(), 2, 1, (C.c_double * 2)(0.0, 0.0), 0, out.data_ptr(), ctx.stream), "x")
    assert out.cpu().tolist() == [0, -1]
    # a one-candidate, one-chunk ant round from the root (1-row history), model and tape dynamics, action_horizon 1 and 2
    from ditreeonlineplanner_amd.engine import AntExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    net = NoisePredNet(input_dim=8, additional_global_cond_dim=97, pred_horizon=16, local_map_size=16, seed=0)
    net.bind(ctx, precision=_lib.PREC_F16X3, max_batch=1)
    rng = np.random.default_rng(0)
    st = np.zeros(29)
    st[:2], st[2], st[3] = [-30.0, -30.0], 0.75, 1.0
    norm = np.concatenate([np.zeros(27), np.ones(27), np.zeros(8), np.ones(8)])
    for A, dyn in ((2, "model"), (1, "tape"), (3, "tape")):
        eng = AntExpansionEngine(ctx, maze, st, st, norm=norm, batch=1, capacity=8, edge_length=A, action_horizon=A, dynamics=dyn)
        obs = rng.normal(size=(1, 1, A, 29))
        obs[..., :2] = [-30.0, -30.0]
        obs[..., 3:7] = [1.0, 0, 0, 0]
        cnt = eng.expand_round(dev(np.zeros((1, 29))), dev(np.zeros((1, 2))), noise=torch.randn(1, 1, 16, 8, device="cuda"),
                               next_obs_tape=dev(obs) if dyn == "tape" else None)
        assert int(cnt[0]) == 2 and torch.isfinite(eng.rb.actions).all() and eng.rb.actions.shape == (1, 1, A, 8)
        ns = int(eng.tree.edge_nstates[1])                     # rows kept (the goal == the start here: the edge may end at step 1)
        n = min(3, ns)
        assert 2 <= ns <= A + 1 and int(eng.tree.hist_n[1]) == n
        assert torch.equal(eng.tree.hist[1, 3 - n:], eng.tree.edge_states[1, ns - n: ns])
