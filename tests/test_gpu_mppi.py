"""GPU: the MPPI controller (`ditree_mppi_step`, facade `MPPI.mppi.MPPI`) -- SURVEY.md 8(f4), BASELINE config 5.

The reference repository does not ship an MPPI module (`run_scenarios_with_lidar_MPPI.py:10` imports one that is absent), so
there is nothing of the reference's to compare with: **parity unpinned**.  The kernels are held to the build's own numpy
restatement (oracle/mppi.py: collision / goal flags exact, costs / controls 1e-9 on the same noise tape) and to invariants
of the algorithm (K = 1 reduces to the nominal sequence, the weights sum to one, the quad-lane and the single-lane kernels
agree bit for bit, a closed loop tracks a collision-free reference path into the goal)."""
import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import mppi as OM
from tests.util import load_maze

pytestmark = pytest.mark.gpu
ROLL, UPD = 1, 1 | 2 | 4 | 8                 # include/ditree.h DITREE_MPPI_*: rollouts; rollouts + min + sums + apply


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def l_path(maze, ds=0.02):
    """East along the bottom corridor of boxes.csv (row 18), then north along column 18: points every `ds`."""
    a = G.cell_rowcol_to_xy([18, 1], maze)
    b = G.cell_rowcol_to_xy([18, 18], maze)
    c = G.cell_rowcol_to_xy([1, 18], maze)
    seg1 = a + (b - a) * np.linspace(0, 1, int(np.linalg.norm(b - a) / ds))[:, None]
    seg2 = b + (c - b) * np.linspace(0, 1, int(np.linalg.norm(c - b) / ds))[1:, None]
    return np.concatenate([seg1, seg2]), c


def make(ctx, K, T=16, **kw):
    from ditreeonlineplanner_amd.mppi import MPPI
    maze = load_maze("boxes")
    path, goal_xy = l_path(maze)
    m = MPPI(maze_data=maze, T=T, K=K, nx=6, nu=2, ctx=ctx, **kw)
    start = np.array([path[0, 0], path[0, 1], 0.0, 0.0, 0.0, 0.0])
    m.reset(start_state=start, goal_state=np.array([goal_xy[0], goal_xy[1], 0, 0, 0, 0.0]))
    m.set_ref_path(path)
    return m, maze, path, start


def kw_of(m):
    p = m.params
    return dict(lam=p.lam, sigma=(p.sigma[0], p.sigma[1]), w_track=p.w_track, w_progress=p.w_progress,
                w_collision=p.w_collision, w_goal=p.w_goal, window_back=p.window_back, window_fwd=p.window_fwd)


def tape(K, T, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(K, T, 2, generator=g, dtype=torch.float64)
    e[..., 0] *= sigma[0]
    e[..., 1] *= sigma[1]
    return e


@pytest.mark.parametrize("lanes", [1, 2, 4])
@pytest.mark.parametrize("case", ["corridor", "near_wall", "near_goal"])
def test_rollout_costs_match_the_numpy_restatement(ctx, lanes, case):
    K, T = 768, 16
    m, maze, path, start = make(ctx, K, T, lanes=lanes)
    if case == "corridor":
        state = np.array([path[300, 0], path[300, 1] + 0.1, 0.2, 2.0, 0.3, 0.05])
    elif case == "near_wall":               # heading at the bottom wall at speed: most noisy rollouts collide
        state = np.array([path[200, 0], path[200, 1] - 0.25, -0.6, 3.0, 0.5, 0.0])
    else:                                   # inside the last cells before the goal: rollouts end inside the goal radius
        state = np.array([path[-40, 0], path[-40, 1], np.pi / 2, 2.5, 0.4, 0.0])
    rng = np.random.default_rng(3)
    U = np.stack([rng.normal(0.5, 2.0, T), rng.normal(0.0, 0.5, T)], axis=1)
    noise = tape(K, T, kw_of(m)["sigma"], 11)
    m._state.copy_(torch.as_tensor(state))
    m._U.copy_(torch.as_tensor(U))
    m.launch(ROLL, noise=noise.to(ctx.device))
    costs = m._costs.cpu().numpy()
    flags = m._flags.cpu().numpy()
    i0 = int(m._result[5].item())
    rc, rf, ri0 = OM.rollout_costs(maze, state, U, path, m.env.goal, noise.numpy(), **kw_of(m))
    assert i0 == ri0
    assert np.array_equal(flags, rf), (np.bincount(flags, minlength=3), np.bincount(rf, minlength=3))
    assert np.abs(costs - rc).max() < 1e-9 * max(1.0, np.abs(rc).max())
    if case == "near_wall":
        assert (rf == 2).sum() > K // 4
    if case == "near_goal":
        assert (rf == 1).sum() > K // 4


def test_quad_and_single_lane_kernels_agree_bit_for_bit(ctx):
    K, T = 4096, 16
    out = []
    for lanes in (1, 2, 4):
        m, maze, path, start = make(ctx, K, T, lanes=lanes, seed=5)
        m._state.copy_(torch.as_tensor(np.array([path[500, 0], path[500, 1] - 0.2, -0.3, 2.5, 0.5, 0.1])))
        m._U.copy_(torch.as_tensor(np.tile(np.array([1.0, 0.2]), (T, 1))))
        m.counter = 9
        m.launch(UPD)                                                 # rollouts + update with device noise
        out.append((m._costs.cpu().numpy(), m._flags.cpu().numpy(), m._U.cpu().numpy(), m._result.cpu().numpy()))
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1])
        assert np.array_equal(out[0][2], o[2]) and np.array_equal(out[0][3][3:8], o[3][3:8])


def test_update_matches_the_restatement_and_weights_sum_to_one(ctx):
    K, T = 1000, 16                                                  # not a multiple of the 256-rollout slices
    m, maze, path, start = make(ctx, K, T)
    state = np.array([path[300, 0], path[300, 1] + 0.1, 0.2, 2.0, 0.3, 0.05])
    rng = np.random.default_rng(4)
    U = np.stack([rng.normal(0.5, 1.0, T), rng.normal(0.0, 0.3, T)], axis=1)
    noise = tape(K, T, kw_of(m)["sigma"], 12)
    m._state.copy_(torch.as_tensor(state))
    m._U.copy_(torch.as_tensor(U))
    w = torch.zeros(K, dtype=torch.float64, device=ctx.device)
    m.launch(UPD, noise=noise.to(ctx.device), weights=w)
    rc, rf, _ = OM.rollout_costs(maze, state, U, path, m.env.goal, noise.numpy(), **kw_of(m))
    Un, wn, beta, eta, ess = OM.update(U, rc, noise.numpy(), kw_of(m)["lam"])
    res = m._result.cpu().numpy()
    wg = w.cpu().numpy()
    assert abs(wg.sum() - 1.0) < 1e-12 and (wg >= 0).all()
    assert np.abs(wg - wn).max() < 1e-9
    assert abs(res[3] - beta) < 1e-9 and abs(res[4] - eta) < 1e-9 * eta and abs(res[7] - ess) < 1e-6 * ess
    assert int(res[6]) == int((rf == 2).sum())
    assert np.abs(m._U.cpu().numpy() - Un).max() < 1e-9
    # a second launch of the same call reproduces the update bit for bit (fixed summation order)
    m._U.copy_(torch.as_tensor(U))
    m.launch(UPD, noise=noise.to(ctx.device))
    assert np.abs(m._U.cpu().numpy() - Un).max() < 1e-9
    U1 = m._U.cpu().numpy().copy()
    m._U.copy_(torch.as_tensor(U))
    m.launch(UPD, noise=noise.to(ctx.device))
    assert np.array_equal(m._U.cpu().numpy(), U1)


def test_one_rollout_reduces_to_the_nominal_sequence(ctx):
    """K = 1: the only rollout is the noise-free nominal sequence, its weight is 1, the controls do not change, and the
    executed step is the env step of the first nominal control."""
    T = 16
    m, maze, path, start = make(ctx, 1, T)
    state = np.array([path[100, 0], path[100, 1], 0.0, 1.5, 0.3, 0.0])
    U = np.tile(np.array([12.0, -0.3]), (T, 1)) * np.linspace(1, 0.5, T)[:, None]      # the first control is clipped by the env
    m._U.copy_(torch.as_tensor(U))
    w = torch.zeros(1, dtype=torch.float64, device=ctx.device)
    m._state.copy_(torch.as_tensor(state))
    m.launch(UPD, weights=w)
    assert w.item() == 1.0 and np.array_equal(m._U.cpu().numpy(), U)
    nxt, action, done = m.step(state)
    x_ref, a_ref, status, U_ref = OM.execute(maze, state, U, m.env.goal)
    assert done is False and status == 0
    assert np.array_equal(action, a_ref) and action[0] == 10.0
    assert np.abs(nxt - x_ref).max() < 1e-12
    assert np.array_equal(m._U.cpu().numpy(), U_ref)                                    # shifted, last control held


def test_device_noise_is_the_documented_counter_hash(ctx):
    """noise == NULL: eps is generated on the device from (seed, counter, k, t); the host mirror of the hash (oracle/mppi.py)
    fed to the restatement gives the same costs (libm differences of the Box-Muller transform: 1e-9), and successive calls
    draw different noise."""
    K, T = 512, 16
    m, maze, path, start = make(ctx, K, T, seed=1234)
    state = np.array([path[300, 0], path[300, 1] + 0.05, 0.1, 2.0, 0.3, 0.0])
    U = np.zeros((T, 2))
    got = []
    for counter in (0, 1):
        m.counter = counter
        m._state.copy_(torch.as_tensor(state))
        m._U.copy_(torch.as_tensor(U))
        m.launch(ROLL)
        got.append(m._costs.cpu().numpy().copy())
        eps = OM.device_noise(1234, counter, K, T, kw_of(m)["sigma"])
        assert abs(eps[1:, :, 0].std() - 3.0) < 0.1 and abs(eps[1:, :, 1].std() - 0.6) < 0.02
        rc, rf, _ = OM.rollout_costs(maze, state, U, path, m.env.goal, eps, **kw_of(m))
        assert np.array_equal(m._flags.cpu().numpy(), rf)
        assert np.abs(got[-1] - rc).max() < 1e-9 * max(1.0, np.abs(rc).max())
    assert got[0][0] == got[1][0] and not np.allclose(got[0][1:], got[1][1:])         # rollout 0 carries no noise


def test_closed_loop_tracks_the_path_into_the_goal(ctx):
    """The driver's loop (run_scenarios_with_lidar_MPPI.py:417-445) on a collision-free L-shaped reference path through
    boxes.csv: the controller reaches the goal radius without a single collision and stays near the path."""
    m, maze, path, start = make(ctx, 2048, 16, seed=7)
    state = start.copy()
    worst = 0.0
    steps = 0
    done = False
    while not m.is_done(state) and steps < 4000:
        nxt, action, done = m.step(state)
        assert done is not None, f"collision at step {steps}, state {state}"
        assert abs(action[0]) <= 10.0 and abs(action[1]) <= 2.0
        state = nxt
        steps += 1
        worst = max(worst, float(np.min(np.hypot(path[:, 0] - state[0], path[:, 1] - state[1]))))
    assert m.is_done(state) and done is True, (steps, state)
    assert worst < 0.45                                           # inside the corridor cell all the way (cells are 1 wide)
    assert steps < 3000                                           # 34 cells at >= 0.6 cells / s
    print("closed loop:", steps, "steps, worst path distance", worst, m.last)


def test_collision_of_the_executed_step_returns_none_and_restarts_the_controls(ctx):
    m, maze, path, start = make(ctx, 64, 16)
    # nose 0.12 from the bottom wall, fast, heading into it: the executed step collides whatever the controls are
    wall_y = G.cell_rowcol_to_xy([18, 5], maze)[1] - 0.5
    state = np.array([path[300, 0], wall_y + 0.19, -np.pi / 2, 5.0, 1.0, 0.0])
    m._U.fill_(1.0)
    nxt, action, done = m.step(state)
    assert done is None and np.array_equal(nxt, state)
    assert (m._U.cpu().numpy() == 0).all()
    with pytest.raises(Exception):
        from ditreeonlineplanner_amd.mppi import MPPI
        MPPI(maze_data=maze, T=100, K=8)


def test_driver_surface(ctx):
    """What run_scenarios_with_lidar_MPPI.py touches on the object: env (lidar, reset_done, set_state, prob_map),
    update_maze, reference_path, set_ref_path thinning."""
    m, maze, path, start = make(ctx, 32, 8)
    assert hasattr(m.env, "lidar2dsim") and m.env.lidar2dsim.scan_time == 0.2 and m.env.prob_map.shape == maze.shape
    assert np.array_equal(m.reference_path, path)
    m.env.reset_done()
    known = maze.copy()
    known[5, 5] = 1
    m.update_maze(known)
    assert m.env.maze_map[5, 5] == 1
    long_path = np.repeat(path, 4, axis=0)
    m.set_ref_path(long_path)
    assert m._path.shape[0] == 4096 and len(m.reference_path) == len(long_path)
    nxt, action, done = m.step(start)
    assert done is False and nxt.shape == (6,)


def test_known_obstacle_bends_the_controls_away_from_the_reference_path(ctx):
    """update_maze: a cell on the reference path becomes occupied in the KNOWN maze (the lidar's finding).  The planning
    rollouts collide there, so the controller stops following the path into it: with the obstacle known it never collides
    (the executed step is checked against the same maze) and keeps its distance, where the controller that does not know it
    drives straight through the cell."""
    maze = load_maze("boxes")
    blocked = maze.copy()
    blocked[18, 9] = 1                                             # on the bottom corridor, 8 cells ahead of the start
    cell_xy = G.cell_rowcol_to_xy([18, 9], maze)
    out = {}
    for name, known in (("unknown", maze), ("known", blocked)):
        m, _, path, start = make(ctx, 2048, 16, seed=11)
        m.update_maze(known)
        state = start.copy()
        closest = np.inf
        for step in range(1500):
            nxt, action, done = m.step(state)
            if done is None:
                break
            state = nxt
            closest = min(closest, float(np.hypot(state[0] - cell_xy[0], state[1] - cell_xy[1])))
            if closest < 0.3 or state[0] > cell_xy[0] + 1.0:
                break
        out[name] = (closest, done, state.copy())
    assert out["unknown"][0] < 0.3                                 # drives through the cell it does not know about
    assert out["known"][1] is not None or out["known"][0] > 0.55   # never inside the occupied cell (half-width 0.5 + ball)
    assert out["known"][0] > 0.55, out["known"]


def test_controller_step_at_config5_size(ctx):
    """BASELINE config 5 as written: ONE controller step at K = 65 536 rollouts, T = 16, on-device noise -- against the numpy
    restatement on the mirrored counter-hash noise: collided / goal flags exact, beta, eta, effective sample size and the
    updated controls 1e-9 (relative where the quantity scales with K)."""
    K, T = 65536, 16
    m, maze, path, start = make(ctx, K, T, seed=4242)
    state = np.array([path[250, 0], path[250, 1] - 0.15, -0.35, 2.8, 0.5, 0.02])       # towards the bottom wall: a share collides
    rng = np.random.default_rng(8)
    U = np.stack([rng.normal(0.5, 1.0, T), rng.normal(0.0, 0.3, T)], axis=1)
    m.counter = 3
    m._state.copy_(torch.as_tensor(state))
    m._U.copy_(torch.as_tensor(U))
    w = torch.zeros(K, dtype=torch.float64, device=ctx.device)
    m.launch(UPD, weights=w)
    kw = kw_of(m)
    eps = OM.device_noise(4242, 3, K, T, kw["sigma"])
    rc, rf, i0 = OM.rollout_costs(maze, state, U, path, m.env.goal, eps, **kw)
    Un, wn, beta, eta, ess = OM.update(U, rc, eps, kw["lam"])
    res = m._result.cpu().numpy()
    flags = m._flags.cpu().numpy()
    assert int(res[5]) == i0
    assert np.array_equal(flags, rf), (np.bincount(flags, minlength=3), np.bincount(rf, minlength=3))
    assert int(res[6]) == int((rf == 2).sum()) and 1000 < int(res[6]) < K - 1000
    assert np.abs(m._costs.cpu().numpy() - rc).max() < 1e-9 * max(1.0, np.abs(rc).max())
    assert abs(res[3] - beta) < 1e-9 * max(1.0, abs(beta)) and abs(res[4] - eta) < 1e-9 * eta and abs(res[7] - ess) < 1e-7 * ess
    assert np.abs(m._U.cpu().numpy() - Un).max() < 1e-9
    wg = w.cpu().numpy()
    assert abs(wg.sum() - 1.0) < 1e-10 and np.abs(wg - wn).max() < 1e-9


# ------------------------------------------------------------------------------------------ config 5 as written: MPPI on the ant slot
def make_ant(ctx, K, T=16, **kw):
    """MPPI(maze, T, K, nx=29, nu=8): the stand-in crawler model (NOT MuJoCo) behind the same controller; boxes.csv scaled by 4."""
    from ditreeonlineplanner_amd.mppi import MPPI
    from oracle import ant as OA
    maze = load_maze("boxes")
    path1, goal_xy = l_path(maze, ds=0.02)
    path = path1 * 4.0                                       # the ant's maze is the car's scaled by s_global = 4
    m = MPPI(maze_data=maze, T=T, K=K, nx=29, nu=8, ctx=ctx, **kw)
    start = np.zeros(29)
    start[:2] = path[0]
    start[2], start[3] = 0.75, 1.0
    start[7:15] = np.tile([0.0, OA.AntModel.ank_rest], 4)
    goal = np.zeros(29)
    goal[:2] = goal_xy * 4.0
    m.reset(start_state=start, goal_state=goal)
    m.set_ref_path(path)
    return m, maze, path, start


def kw_ant(m):
    p = m.params
    return dict(lam=p.lam, sigma=[p.sigma[d] for d in range(8)], w_track=p.w_track, w_progress=p.w_progress, w_collision=p.w_collision,
                w_goal=p.w_goal, window_back=p.window_back, window_fwd=p.window_fwd)


def test_ant_mppi_matches_the_numpy_restatement(ctx):
    """ditree_mppi_step_ant at K = 4096, T = 16 on the on-device noise: costs / beta / eta / controls 1e-9 against oracle/mppi.py
    (rollout_costs_ant on the mirrored counter hash), collided / goal flags exact; K = 1 reduces to the nominal sequence and the
    executed step is the model step of its first control."""
    from ditreeonlineplanner_amd.mppi import AntMPPI
    K, T = 4096, 16
    m, maze, path, start = make_ant(ctx, K, T, seed=77)
    assert isinstance(m, AntMPPI)
    state = start.copy()
    state[:2] = path[400] + np.array([0.0, -0.7])            # 0.1 from touching the bottom wall (ball radius 1.2, cell 4), drifting
    state[15:18] = [0.8, -1.2, 0.0]                          # towards it: most noisy rollouts collide, some do not
    rng = np.random.default_rng(2)
    U = rng.uniform(-0.5, 0.5, (T, 8))
    m.counter = 5
    m._state.copy_(torch.as_tensor(state))
    m._U.copy_(torch.as_tensor(U))
    w = torch.zeros(K, dtype=torch.float64, device=ctx.device)
    m.launch(UPD, weights=w)
    kw = kw_ant(m)
    eps = OM.device_noise_ant(77, 5, 0, K, T, kw["sigma"])
    assert abs(eps[1:].std() - 0.5) < 0.01
    rc, rf, i0 = OM.rollout_costs_ant(maze, state, U, path, m.env.goal, eps, **kw)
    Un, wn, beta, eta, ess = OM.update_ant(U, rc, eps, kw["lam"])
    res = m._result.cpu().numpy()
    assert int(res[5]) == i0
    flags = m._flags.cpu().numpy()
    assert np.array_equal(flags, rf), (np.bincount(flags, minlength=3), np.bincount(rf, minlength=3))
    assert 50 < int((rf == 2).sum()) < K - 50 and int(res[6]) == int((rf == 2).sum())
    assert np.abs(m._costs.cpu().numpy() - rc).max() < 1e-9 * max(1.0, np.abs(rc).max())
    assert abs(res[3] - beta) < 1e-9 * max(1.0, abs(beta)) and abs(res[4] - eta) < 1e-9 * eta and abs(res[7] - ess) < 1e-7 * ess
    assert np.abs(m._U.cpu().numpy() - Un).max() < 1e-9
    wg = w.cpu().numpy()
    assert abs(wg.sum() - 1.0) < 1e-10 and np.abs(wg - wn).max() < 1e-9
    # K = 1: the nominal sequence alone; the executed step = one model step of the (clipped) first control
    m1, _, _, _ = make_ant(ctx, 1, T)
    U1 = rng.uniform(-1.4, 1.4, (T, 8))
    m1._U.copy_(torch.as_tensor(U1))
    nxt, action, done = m1.step(start)
    x_ref, a_ref, status, U_ref = OM.execute_ant(maze, start, U1, m1.env.goal)
    assert done is False and status == 0 and np.array_equal(action, a_ref) and np.abs(action).max() == 1.0
    assert np.abs(nxt - x_ref).max() < 1e-9 and np.array_equal(m1._U.cpu().numpy(), U_ref)
    # closed loop for a few steps: finite, no collision from the corridor centre, the state the facade hands back is the device's
    m2, _, _, _ = make_ant(ctx, 2048, T, seed=3)
    st = start.copy()
    for _ in range(20):
        st, a, done = m2.step(st)
        assert done is False and np.isfinite(st).all() and np.abs(a).max() <= 1.0
    assert m2.last["effective_samples"] > 1.0
