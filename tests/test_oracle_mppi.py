"""CPU: self-consistency of the MPPI restatement (oracle/mppi.py).  There is no reference MPPI module and therefore no golden
vector: PARITY UNPINNED; these are invariants of the algorithm the HIP kernels are then compared with (tests/test_gpu_mppi.py)."""
import numpy as np

from oracle import geometry as G
from oracle import mppi as OM
from tests.util import load_maze

KW = dict(lam=1.0, sigma=(3.0, 0.6), w_track=20.0, w_progress=0.5, w_collision=1e3, w_goal=50.0, window_back=8, window_fwd=56)


def _setup():
    maze = load_maze("boxes")
    a, b = G.cell_rowcol_to_xy([18, 1], maze), G.cell_rowcol_to_xy([18, 18], maze)
    path = a + (b - a) * np.linspace(0, 1, 850)[:, None]
    return maze, path, b


def test_noise_free_rollout_and_weights():
    maze, path, goal = _setup()
    state = np.array([path[100, 0], path[100, 1], 0.0, 2.0, 0.3, 0.0])
    T, K = 16, 64
    U = np.tile(np.array([1.0, 0.0]), (T, 1))
    eps = OM.device_noise(1, 0, K, T, KW["sigma"])
    c, f, i0 = OM.rollout_costs(maze, state, U, path, goal, eps, **KW)
    assert i0 == 100 and f[0] == 0
    # rollout 0 ignores its noise row: the same cost with any tape
    c2, _, _ = OM.rollout_costs(maze, state, U, path, goal, eps * 0.0, **KW)
    assert c[0] == c2[0] and (c2 == c2[0]).all()
    Un, w, beta, eta, ess = OM.update(U, c, eps, KW["lam"])
    assert abs(w.sum() - 1) < 1e-12 and beta == c.min() and 1.0 <= ess <= K
    # one rollout: nothing to average, the controls stay
    U1, w1, _, _, _ = OM.update(U, c[:1], eps[:1], KW["lam"])
    assert np.array_equal(U1, U) and w1[0] == 1.0


def test_collision_and_goal_end_a_rollout():
    maze, path, goal = _setup()
    T = 16
    U = np.zeros((T, 2))
    wall_y = G.cell_rowcol_to_xy([18, 5], maze)[1] - 0.5
    into_wall = np.array([path[300, 0], wall_y + 0.3, -np.pi / 2, 5.0, 1.0, 0.0])
    c, f, _ = OM.rollout_costs(maze, into_wall, U, path, goal, np.zeros((2, T, 2)), **KW)
    assert (f == 2).all() and (c > KW["w_collision"]).all()
    at_goal = np.array([goal[0] - 0.6, goal[1], 0.0, 3.0, 0.5, 0.0])
    c, f, _ = OM.rollout_costs(maze, at_goal, U, path, goal, np.zeros((2, T, 2)), **KW)
    assert (f == 1).all() and (c < 0).all()
    x, a, status, Un = OM.execute(maze, into_wall.copy(), np.ones((T, 2)), goal)
    assert status == 0 or status == 2
    x, a, status, Un = OM.execute(maze, np.array([path[300, 0], wall_y + 0.19, -np.pi / 2, 5.0, 1.0, 0.0]), np.ones((T, 2)), goal)
    assert status == 2 and (Un == 0).all()


def test_device_noise_hash_statistics_and_determinism():
    e = OM.device_noise(9, 4, 4096, 16, (3.0, 0.6))
    assert np.array_equal(e, OM.device_noise(9, 4, 4096, 16, (3.0, 0.6)))
    assert not np.allclose(e, OM.device_noise(9, 5, 4096, 16, (3.0, 0.6)))
    assert abs(e[..., 0].std() - 3.0) < 0.05 and abs(e[..., 1].std() - 0.6) < 0.01 and np.abs(e.mean(axis=(0, 1))).max() < 0.05
    assert abs(np.corrcoef(e[..., 0].ravel(), e[..., 1].ravel())[0, 1]) < 0.02
