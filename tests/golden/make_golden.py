"""Generate the golden vectors under tests/golden/ by importing the reference's modules.

Runs ONLY in the build container (needs /root/reference); nothing here travels to the
GPU box except the .npz/.json files it writes.  The reference is imported unmodified;
third-party imports it never calls on this path (spatialmath, termcolor, minari,
diffusers) are satisfied with empty in-memory placeholder modules.  ``car_env`` (casadi,
gymnasium) and ``local_map_encoder`` (torchvision) cannot be imported here, so the
dynamics and the ResNet encoder have no executable reference: see oracle/__init__.py.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)


def _placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod("spatialmath")
    mod("spatialmath.base", r2q=None)
    mod("spatialmath.base.transforms3d", isrot=None)
    mod("termcolor", cprint=print)
    mod("minari")
    mod("diffusers")
    mod("diffusers.schedulers")
    mod("diffusers.schedulers.scheduling_ddpm", DDPMScheduler=object)


_placeholders()
import common.map_utils as ref_mu                                   # noqa: E402
from common.fm_utils import get_timesteps as ref_get_timesteps      # noqa: E402
from lidar_sim.lidar_2d_sim import Lidar2DSim as RefLidar           # noqa: E402
from model.diffusion.conditional_unet1d import ConditionalUnet1D as RefUnet   # noqa: E402
import policies.fm_policy as ref_fm                                  # noqa: E402
import planners.RRT as ref_rrt                                       # noqa: E402
import planners.base_planner as ref_bp                               # noqa: E402

from oracle import geometry as G                                     # noqa: E402
from oracle import rrt as ORRT                                       # noqa: E402
from oracle import sampler as OS                                     # noqa: E402
from oracle import denoiser as OD                                    # noqa: E402
from oracle.tapes import ActionTape                                  # noqa: E402

MAZE_DIR = os.path.join(REF, "maps", "mazes")


def load_maze(name):
    return np.loadtxt(os.path.join(MAZE_DIR, f"{name}.csv"), delimiter=",")


def random_poses(rng, maze, n, boundary_frac=0.25, oob_frac=0.05):
    H, W = maze.shape
    x = rng.uniform(-W / 2, W / 2, n)
    y = rng.uniform(-H / 2, H / 2, n)
    th = rng.uniform(-np.pi, np.pi, n)
    nb = int(n * boundary_frac)
    # snap a fraction onto / next to cell boundaries
    x[:nb] = np.round(x[:nb]) + rng.choice([0.0, 1e-12, -1e-12, 0.1, -0.1, 0.025], nb)
    y[nb:2 * nb] = np.round(y[nb:2 * nb]) + rng.choice([0.0, 1e-12, -1e-12, 0.1, -0.1, 0.025], nb)
    no = int(n * oob_frac)
    if no > 0:
        x[-no:] = rng.uniform(-W / 2 - 1, W / 2 + 1, no)
        y[-no:] = rng.uniform(-H / 2 - 1, H / 2 + 1, no)
    return np.stack([x, y, th], axis=1)


def gen_collision(out):
    rng = np.random.default_rng(101)
    for name in ("Race_Track", "boxes", "random_huge", "narrow_short"):
        maze = load_maze(name)
        poses = random_poses(rng, maze, 4096)
        exp = np.array([bool(ref_mu.is_colliding_car(p, maze)) for p in poses])
        mine = G.is_colliding_car(poses, maze)
        assert (exp == mine).all(), f"oracle collision mismatch on {name}: {np.nonzero(exp != mine)[0][:10]}"
        out[f"collision_{name}_poses"] = poses
        out[f"collision_{name}_expected"] = exp
        print(f"collision {name}: {exp.mean():.3f} colliding, oracle == reference")


def gen_local_map(out):
    rng = np.random.default_rng(102)
    for name in ("Race_Track", "boxes", "random_huge"):
        maze = np.float32(load_maze(name))
        H, W = maze.shape
        poses = random_poses(rng, maze, 256, boundary_frac=0.1, oob_frac=0.0)
        center = (W / 2, H / 2)
        for tag, (n, scale, sg) in {"car": (20, 0.2, 1.0), "ant": (16, 0.8, 4.0)}.items():
            exp = np.concatenate([ref_mu.create_local_map(maze, p[0] * sg, p[1] * sg, p[2] if tag == "car" else 0.0,
                                                          n, scale, sg, (center[0] * sg, center[1] * sg))
                                  for p in poses])
            th = poses[:, 2] if tag == "car" else np.zeros(len(poses))
            mine = G.create_local_map(maze, poses[:, 0] * sg, poses[:, 1] * sg, th, n, scale, sg,
                                      (center[0] * sg, center[1] * sg))
            assert exp.dtype == np.float32 and (exp == mine).all(), f"local map mismatch {name} {tag}"
            out[f"localmap_{name}_{tag}_poses"] = poses
            out[f"localmap_{name}_{tag}_expected"] = np.packbits(exp.astype(np.uint8), axis=None)
        print(f"local map {name}: oracle == reference")


class _RecordingNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, sample, local_map, timestep, global_cond):
        self.calls.append(dict(sample=sample.clone(), local_map=local_map.clone(),
                               timestep=timestep.clone(), cond=global_cond.clone()))
        return 0.25 * sample + 0.5          # any deterministic velocity


def _ref_sampler(net, k_iters=1):
    """DiffusionSampler without its __init__ (which unpickles metadata/*.pt with
    weights_only=False -- not allowed here); attributes set as run_scenarios.py:179-185."""
    s = ref_fm.DiffusionSampler.__new__(ref_fm.DiffusionSampler)
    torch.nn.Module.__init__(s)
    s.device = "cpu"
    s.metadata = {k: v.copy() for k, v in OS.CAR_META.items()}
    s.action_dim, s.prediction_type, s.env_id, s.policy = 2, "actions", "carmaze", "flow_matching"
    s.num_diffusion_iters, s.pred_horizon, s.obs_history, s.action_history = k_iters, 64, 1, 1
    s.position_conditioned, s.goal_conditioned, s.local_map_conditioned = False, True, True
    s.local_map_size = 20
    s.noise_pred_net, s.noise_scheduler = net, None
    return s


def gen_sampler(out):
    rng = np.random.default_rng(103)
    maze = np.float32(load_maze("boxes"))
    n = 64
    states = np.stack([rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), rng.uniform(-np.pi, np.pi, n),
                       rng.uniform(-1, 5, n), rng.uniform(-1, 1, n), rng.uniform(-0.4, 0.4, n)], axis=1)
    prev = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n)], axis=1)
    has_prev = rng.random(n) < 0.7
    goals = np.stack([rng.uniform(-9, 9, n), rng.uniform(-9, 9, n)], axis=1)
    conds, acts, noises, maps = [], [], [], []
    for i in range(n):
        net = _RecordingNet()
        smp = _ref_sampler(net)
        lm = ref_mu.create_local_map(maze, states[i, 0], states[i, 1], states[i, 2], 20, 0.2, 1.0, (10.0, 10.0))
        torch.manual_seed(1000 + i)
        # history of 3 observations: only the last one may matter (obs_history = 1)
        hist = np.stack([states[(i + 1) % n], states[(i + 2) % n], states[i]])[None]
        pa = np.stack([prev[(i + 1) % n], prev[i]])[None] if has_prev[i] else None
        a = smp(hist, prev_actions=pa, goal=goals[i], local_map=torch.tensor(lm))
        c = net.calls[0]
        assert float(c["timestep"][0]) == 0.0 and len(net.calls) == 1
        assert (c["local_map"].numpy() == OS.scale_local_map(lm)).all()
        conds.append(c["cond"].numpy()[0])
        noises.append(c["sample"].numpy()[0])
        acts.append(a[0])
        maps.append(lm[0])
    conds, acts, noises = np.array(conds), np.array(acts), np.array(noises)
    mine = OS.car_cond_vector(states, prev, has_prev, goals)
    err = np.abs(mine - conds).max()
    assert err < 2e-7, err
    x1 = noises + (0.25 * noises + 0.5) * 1.0
    assert np.abs(OS.unnormalize_actions(x1.astype(np.float32)) - acts).max() < 1e-6
    out.update(sampler_states=states, sampler_prev=prev, sampler_has_prev=has_prev, sampler_goals=goals,
               sampler_cond_expected=conds, sampler_noise=noises, sampler_actions_expected=acts)
    print(f"sampler pre/post: oracle vs reference max |d cond| = {err:.2e}")


def gen_sampler_ant(out):
    """The antmaze branch of the reference's DiffusionSampler.forward (quaternion -> rot6d of the normalised values, 3-step
    history with front padding, 8-d previous action, yaw = 0) on a recording net; 1-, 2- and 3-step histories, with and
    without a previous action."""
    rng = np.random.default_rng(107)
    n = 48
    obs = rng.normal(0.0, 1.0, (n, 3, 29))
    obs[..., :2] = rng.uniform(-18, 18, (n, 3, 2))
    obs[..., 2] = rng.uniform(0.3, 0.9, (n, 3))
    q = rng.normal(size=(n, 3, 4))
    obs[..., 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    prev = rng.uniform(-1, 1, (n, 8))
    has_prev = rng.random(n) < 0.7
    n_hist = rng.integers(1, 4, n)
    goals = rng.uniform(-18, 18, (n, 2))
    conds, acts = [], []
    for i in range(n):
        net = _RecordingNet()
        smp = _ref_sampler(net)
        smp.metadata = {k: v.copy() for k, v in OS.ANT_META.items()}
        smp.action_dim, smp.env_id, smp.pred_horizon, smp.obs_history, smp.local_map_size = 8, "antmaze", 16, 3, 16
        hist = obs[i, 3 - n_hist[i]:][None]
        pa = prev[i][None, None] if has_prev[i] else None
        torch.manual_seed(2000 + i)
        a = smp(hist, prev_actions=pa, goal=goals[i], local_map=torch.zeros(1, 16, 16))
        c = net.calls[0]
        assert c["sample"].shape == (1, 16, 8) and c["cond"].shape == (1, 97)
        conds.append(c["cond"].numpy()[0])
        acts.append(a[0])
        mine = OS.ant_cond_vector(hist, prev[i][None], has_prev[i:i + 1], goals[i])
        assert np.abs(mine - conds[-1]).max() < 2e-7, (i, np.abs(mine - conds[-1]).max())
    out.update(sampler_ant_obs=obs, sampler_ant_n_hist=n_hist, sampler_ant_prev=prev, sampler_ant_has_prev=has_prev,
               sampler_ant_goals=goals, sampler_ant_cond_expected=np.array(conds))
    print("ant sampler pre-processing: oracle == reference on", n, "cases")


def gen_timesteps(out_json):
    res = {}
    for k in (1, 2, 4, 5, 8, 16):
        t0, dt = ref_get_timesteps("exp", k, 4.0)
        m0, md = OS.get_timesteps("exp", k, 4.0)
        assert torch.equal(t0, m0) and torch.equal(dt, md)
        res[str(k)] = {"t0": [float(v) for v in t0], "dt": [float(v) for v in dt]}
    out_json["get_timesteps_exp_4"] = res
    print("get_timesteps: oracle == reference")


def gen_unet(out):
    for tag, (inp, gcd, P) in {"car": (2, 407, 64), "ant": (8, 497, 16)}.items():
        torch.manual_seed(7)
        ref = RefUnet(input_dim=inp, global_cond_dim=gcd, down_dims=[512, 1024, 2048]).eval()
        torch.manual_seed(7)
        mine = OD.OracleUnet1D(inp, gcd).eval()
        rs, ms = ref.state_dict(), mine.state_dict()
        assert set(rs) == set(ms), set(rs) ^ set(ms)
        for k in rs:
            assert torch.equal(rs[k], ms[k]), k
        g = torch.Generator().manual_seed(11)
        B = 4
        x = torch.randn(B, P, inp, generator=g)
        cond = torch.randn(B, gcd, generator=g) * 0.5
        ts = torch.tensor([0.0, 0.0, 3.7, 12.5])
        with torch.no_grad():
            y_ref = ref(x, ts, global_cond=cond)
            y_mine = mine(x, ts, cond)
        err = (y_ref - y_mine).abs().max().item()
        assert err < 1e-5, err
        n_par = sum(p.numel() for p in ref.parameters())
        cks = {k: float(v.double().sum()) for k, v in list(rs.items())[:8]}
        out[f"unet_{tag}_x"] = x.numpy()
        out[f"unet_{tag}_cond"] = cond.numpy()
        out[f"unet_{tag}_t"] = ts.numpy()
        out[f"unet_{tag}_y_expected"] = y_ref.numpy()
        out[f"unet_{tag}_nparams"] = np.array(n_par)
        out[f"unet_{tag}_wsum"] = np.array(sum(float(v.double().sum()) for v in rs.values()))
        print(f"unet {tag}: {n_par} params, seeded weights identical, max |dy| = {err:.2e}", cks and "")
        del ref, mine


def gen_lidar(out):
    rng = np.random.default_rng(104)
    maze = load_maze("boxes")
    lid = RefLidar()
    assert (lid.angles_deg == G.LIDAR_ANGLES_DEG).all()
    free = np.argwhere(maze == 0)
    pick = free[rng.choice(len(free), 64, replace=False)]
    poses = np.stack([pick[:, 1] + rng.uniform(0.05, 0.95, 64), pick[:, 0] + rng.uniform(0.05, 0.95, 64),
                      rng.uniform(-np.pi, np.pi, 64)], axis=1)
    D, E, V, Hh = [], [], [], []
    for p in poses:
        d, e, v = lid.scan(p, maze)
        md, me, mv, mh = G.lidar_scan(p, maze)
        assert np.array_equal(d, md) and np.array_equal(e, me) and np.array_equal(v, mv)
        D.append(d), E.append(e), Hh.append(mh)
        vis = np.zeros(maze.shape, dtype=np.uint8)
        vis[v[:, 1], v[:, 0]] = 1
        V.append(vis)
    out.update(lidar_boxes_poses=poses, lidar_boxes_dist=np.array(D), lidar_boxes_end=np.array(E),
               lidar_boxes_visited=np.packbits(np.array(V), axis=None), lidar_boxes_hit=np.array(Hh))
    # the reference's own demo map, lidar_2d_sim.py:137-148
    demo = np.zeros((100, 100))
    demo[30:70, 40] = 1
    demo[50, 20:80] = 1
    demo[10:20, 10:20] = 1
    p = np.array([10.0, 80.0, 90.0])
    d, e, v = lid.scan(p, demo)
    md, me, mv, _ = G.lidar_scan(p, demo)
    assert np.array_equal(d, md) and np.array_equal(e, me) and np.array_equal(v, mv)
    out.update(lidar_demo_dist=d, lidar_demo_end=e)
    print("lidar: oracle == reference (64 poses on boxes + demo map)")


def gen_kdtree(out):
    from scipy.spatial import KDTree
    rng = np.random.default_rng(105)
    for n in (1, 17, 1000):
        nodes = rng.uniform(-10, 10, (n, 2))
        q = rng.uniform(-10, 10, (1024, 2))
        _, idx = KDTree(nodes).query(q, k=1)
        mine = G.nn_argmin(q, nodes)
        assert (idx == mine).all()
        out[f"kdtree_{n}_nodes"] = nodes
        out[f"kdtree_{n}_queries"] = q
        out[f"kdtree_{n}_expected"] = idx.astype(np.int32)
    print("kd-tree nearest: oracle == scipy KDTree")


def gen_dynamics(out):
    """Independent FP64 evaluation of car_env.py:376-390 written with python floats and
    math.* (not the numpy oracle).  Self-referential: car_env cannot execute here."""
    import math
    rng = np.random.default_rng(106)

    def step(s, a):
        x, y, psi, v, D, dl = s
        a0 = min(max(a[0], -10.0), 10.0)
        a1 = min(max(a[1], -2.0), 2.0)
        Fxd = (0.28 - 0.05 * v) * D - 0.006 * (v ** 2) - 0.011 * math.tanh(5.0 * v)
        dot = [v * math.cos(psi + 0.5 * dl), v * math.sin(psi + 0.5 * dl), v * 15.5 * dl,
               (Fxd / 0.043) * math.cos(0.5 * dl), a0, a1]
        return [s[i] + (1.0 / 50.0) * dot[i] for i in range(6)]

    S0 = np.stack([rng.uniform(-5, 5, 32), rng.uniform(-5, 5, 32), rng.uniform(-np.pi, np.pi, 32),
                   rng.uniform(-2, 5, 32), rng.uniform(-1, 1, 32), rng.uniform(-0.4, 0.4, 32)], axis=1)
    S0[0, 3:] = 0.0                      # v = 0: only D, delta change
    S0[1, 5] = 0.0                       # delta = 0: straight line
    Aseq = np.stack([rng.uniform(-12, 12, (32, 64)), rng.uniform(-3, 3, (32, 64))], axis=2)
    traj = np.zeros((32, 65, 6))
    for b in range(32):
        s = list(S0[b])
        traj[b, 0] = s
        for i in range(64):
            s = step(s, Aseq[b, i])
            traj[b, i + 1] = s
    mine = np.zeros_like(traj)
    cur = S0.copy()
    mine[:, 0] = cur
    for i in range(64):
        cur = G.car_step(cur, Aseq[:, i])
        mine[:, i + 1] = cur
    err = np.abs(mine - traj).max()
    assert err < 1e-9, err
    # closed forms
    s1 = G.car_step(S0[0], Aseq[0, 0])
    assert s1[0] == S0[0, 0] and s1[1] == S0[0, 1] and s1[2] == S0[0, 2] and s1[3] == 0.0
    out.update(dyn_s0=S0, dyn_actions=Aseq, dyn_traj_expected=traj)
    print(f"dynamics known-answer: oracle vs independent evaluation max err {err:.2e} (self-referential)")


def gen_dynamics_from_text(out):
    """CarEnv._update_state / _check_done / step taken from the TEXT of the reference's car_env.py with `ast` (the module
    itself cannot be imported: gymnasium, casadi) and executed with numpy; `MX.tanh` -- casadi's tanh applied to a numpy
    float -- is bound to math.tanh.  The physical constants are the `self.X = <literal>` assignments of __init__, the action
    bounds the `model.dthrottle_min ...` literals of bicycle_model().  This pins the EXPRESSION ORDER and the step / freeze
    logic of the reference's own source text; it does not pin casadi's tanh (last-ulp), so a12 stays "parity unpinned"."""
    import ast
    import math
    path = os.path.join(REF, "car_env.py")
    tree = ast.parse(open(path).read(), filename=path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CarEnv"][0]
    keep = {"_update_state", "_check_done", "step", "_calculate_reward", "_get_obs"}
    funcs = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in keep]
    ns = {"np": np, "MX": types.SimpleNamespace(tanh=math.tanh), "is_colliding_car": lambda st, mz: False}
    exec(compile(ast.Module(body=funcs, type_ignores=[]), path, "exec"), ns)
    consts = {}
    init = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__"][0]
    for node in ast.walk(init):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Attribute) \
                and isinstance(node.targets[0].value, ast.Name) and node.targets[0].value.id == "self":
            try:
                consts[node.targets[0].attr] = ast.literal_eval(node.value)
            except (ValueError, SyntaxError):
                if ast.unparse(node.value).replace(" ", "") == "1.0/50.0":
                    consts[node.targets[0].attr] = 1.0 / 50.0
    bounds = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Attribute) \
                and node.targets[0].attr in ("dthrottle_min", "dthrottle_max", "ddelta_min", "ddelta_max"):
            bounds[node.targets[0].attr] = float(ast.literal_eval(node.value))
    assert {"m", "C1", "C2", "Cm1", "Cm2", "Cr0", "Cr2", "dt"} <= set(consts), sorted(consts)
    assert len(bounds) == 4, bounds

    class Env:
        pass
    env = Env()
    for k in ("m", "C1", "C2", "Cm1", "Cm2", "Cr0", "Cr2", "dt"):
        setattr(env, k, consts[k])
    env.action_space = types.SimpleNamespace(low=np.array([bounds["dthrottle_min"], bounds["ddelta_min"]], dtype=np.float32),
                                             high=np.array([bounds["dthrottle_max"], bounds["ddelta_max"]], dtype=np.float32))
    env.collision_checking = False
    for name in keep:
        setattr(Env, name, ns[name])
    rng = np.random.default_rng(108)
    n, T = 48, 64
    S0 = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(-np.pi, np.pi, n),
                   rng.uniform(-2, 5, n), rng.uniform(-1, 1, n), rng.uniform(-0.4, 0.4, n)], axis=1)
    Aseq = np.stack([rng.uniform(-12, 12, (n, T)), rng.uniform(-3, 3, (n, T))], axis=2)
    goals = S0[:, :2] + rng.uniform(-1.5, 1.5, (n, 2))          # near enough that some runs reach the 0.5 m radius and freeze
    traj = np.zeros((n, T + 1, 6))
    succ = np.zeros((n, T), dtype=bool)
    for b in range(n):
        env._state = S0[b].copy()
        env.goal = goals[b].copy()
        env.done = env.terminated = False
        env.current_step = 0
        traj[b, 0] = S0[b]
        for i in range(T):
            obs, rew, term, trunc, info = env.step(Aseq[b, i])
            traj[b, i + 1] = env._state
            succ[b, i] = bool(info["success"])
    # the oracle: one Euler step + goal test, frozen once done (car_env.py:254)
    mine = np.zeros_like(traj)
    msucc = np.zeros_like(succ)
    for b in range(n):
        cur, done = S0[b].copy(), False
        mine[b, 0] = cur
        for i in range(T):
            if not done:
                cur = G.car_step(cur[None], Aseq[b, i][None])[0]
                done = bool(G.goal_reached(cur, goals[b]))
            mine[b, i + 1] = cur
            msucc[b, i] = done
    err = np.abs(mine - traj).max()
    assert np.array_equal(msucc, succ), "goal / freeze logic differs from the reference text"
    assert err == 0.0, err                       # same expressions, same numpy / libm: bit for bit
    assert succ.any() and not succ.all()
    out.update(dyntext_s0=S0, dyntext_actions=Aseq, dyntext_goals=goals, dyntext_traj_expected=traj, dyntext_success=succ)
    print(f"dynamics from the reference's source text ({n} runs x {T} steps, {int(succ[:, -1].sum())} reach the goal and freeze): "
          f"oracle == text, bit for bit")


# --------------------------------------------------------------------------- planner traces
class _TapeSampler(torch.nn.Module):
    """nn.Module so that BasePlanner keeps it (base_planner.py:56-57)."""

    def __init__(self, tape: ActionTape, counter):
        super().__init__()
        self.tape, self.counter = tape, counter
        self.chunk = 0
        self.last_cand = -1

    def forward(self, prev_states, prev_actions=None, goal=None, local_map=None):
        cand = self.counter["cand"] - 1
        if cand != self.last_cand:
            self.chunk, self.last_cand = 0, cand
        a = self.tape.actions(np.array([cand]), self.chunk)
        self.chunk += 1
        return a


def run_reference_planner(scn, n_candidates, tape_seed, prop_duration=(64,)):
    name, maze_name, sr, sc, sdeg, gr, gc = scn
    maze = load_maze(maze_name)
    env = ORRT.OracleCarEnv(maze_map=maze, collision_checking=False)
    start_xy = env.cell_rowcol_to_xy(np.array([sr, sc]))
    goal_xy = env.cell_rowcol_to_xy(np.array([gr, gc]))
    start = np.array([start_xy[0], start_xy[1], np.deg2rad(float(sdeg)), 0.0, 0.0, 0.0])
    goal = np.array([goal_xy[0], goal_xy[1], 0.0, 0.0, 0.0, 0.0])
    counter = {"cand": 0}
    tape = ActionTape(tape_seed)
    smp = _TapeSampler(tape, counter)
    import random
    random.seed(42)
    np.random.seed(42)
    planner = ref_rrt.RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=smp,
                                  prediction_type="actions", action_horizon=8, local_map_size=20,
                                  local_map_scale=0.2, global_map_scale=1.0, goal_conditioning_bias=0.85,
                                  prop_duration=list(prop_duration), time_budget=n_candidates, max_iter=300, verbose=False)
    planner.device = "cpu"
    orig_sample = planner.random_node_sample

    def counted_sample(*a, **k):
        counter["cand"] += 1
        return orig_sample(*a, **k)
    planner.random_node_sample = counted_sample
    real_time = ref_rrt.time.time
    ref_rrt.time.time = lambda: float(counter["cand"])       # budget = number of candidates
    try:
        planner.reset()
        # planner.reset re-seeds nothing; seed again so the tape starts at the plan
        random.seed(42)
        np.random.seed(42)
        path, actions = planner.plan()
    finally:
        ref_rrt.time.time = real_time
    nodes = planner.node_list
    index = {id(n): i for i, n in enumerate(nodes)}
    parents = np.array([-1 if n.parent is None else index[id(n.parent)] for n in nodes], dtype=np.int32)
    states = np.array([n.state for n in nodes])
    reached = bool(env.done) and path is not None and bool(G.goal_reached(states[-1], env.goal))
    return dict(maze=maze, start=start, goal=goal, parents=parents, states=states, reached=reached,
                path=path, actions=actions, iterations=planner.results["iterations"],
                candidates=counter["cand"])


def _ref_planner(maze, start, goal, tape_seed, budget, run_type, counter):
    env = ORRT.OracleCarEnv(maze_map=maze, collision_checking=False, run_type=run_type)
    smp = _TapeSampler(ActionTape(tape_seed), counter)
    planner = ref_rrt.RRT_Planner(start, goal, env_id="carmaze", environment=env, sampler=smp,
                                  prediction_type="actions", action_horizon=8, local_map_size=20,
                                  local_map_scale=0.2, global_map_scale=1.0, goal_conditioning_bias=0.85,
                                  prop_duration=[64], time_budget=budget, max_iter=300, verbose=False,
                                  run_type=run_type)
    planner.device = "cpu"
    orig_sample = planner.random_node_sample
    orig_nearest = planner.nearest_node

    def counted_nearest(sample):                       # one call per candidate whatever the sample source
        counter["cand"] += 1
        return orig_nearest(sample)
    planner.nearest_node = counted_nearest
    return planner, env


def _plan_with_candidate_clock(planner, counter):
    import random
    real_time = ref_rrt.time.time
    ref_rrt.time.time = lambda: float(counter["cand"])
    try:
        random.seed(42)
        np.random.seed(42)
        counter["cand"] = 0
        return planner.plan()
    finally:
        ref_rrt.time.time = real_time


def _tree_of(planner):
    nodes = planner.node_list
    index = {id(n): i for i, n in enumerate(nodes)}
    parents = np.array([-1 if n.parent is None else index[id(n.parent)] for n in nodes], dtype=np.int32)
    return parents, np.array([n.state for n in nodes])


def gen_traces_run_type1(out):
    """run_type 1 ("Original+Ref"): obstacle-ahead flags, reference-path sampling after an obstacle was
    scanned onto the main path, furthest-along-path fallback (RRT.py:61-111,134-140,153-156,202-254)."""
    maze = load_maze("boxes")
    env0 = ORRT.OracleCarEnv(maze_map=maze)
    start = np.array([*env0.cell_rowcol_to_xy(np.array([17, 2])), np.deg2rad(45.0), 0.0, 0.0, 0.0])
    goal = np.array([*env0.cell_rowcol_to_xy(np.array([2, 17])), 0.0, 0.0, 0.0, 0.0])
    seed, n1, n2 = 77, 300, 300
    counter = {"cand": 0}
    planner, env = _ref_planner(maze.copy(), start, goal, seed, n1, 1, counter)
    planner.reset()
    path1, act1 = _plan_with_candidate_clock(planner, counter)
    par1, st1 = _tree_of(planner)
    # obstacle-ahead known answers on the stage-1 tree and on random poses
    rng = np.random.default_rng(107)
    poses = np.concatenate([st1[:, :3], random_poses(rng, maze, 512, oob_frac=0.0)])
    poses = poses[np.abs(poses[:, 0]) < 9.9]
    poses = poses[np.abs(poses[:, 1]) < 9.9]
    ahead = np.array([bool(planner.check_obstacle_ahead(np.concatenate([p, np.zeros(3)]))) for p in poses])
    mine = G.check_obstacle_ahead(np.concatenate([poses, np.zeros((len(poses), 3))], axis=1), maze)
    assert np.array_equal(ahead, mine)
    out["rt1_ahead_poses"] = poses
    out["rt1_ahead_expected"] = ahead
    pl = ORRT.OraclePlanner(maze, start, goal, ActionTape(seed).sampler(), run_type=1)
    r1, opath1, oact1 = pl.plan(ORRT.RandomTape(42), n1, batch=1)
    assert not pl.sticky_triggered
    assert np.array_equal(np.array(pl.tree.parents, dtype=np.int32), par1) and np.array_equal(np.array(pl.tree.states), st1)
    assert (path1 is None) == (opath1 is None)
    assert path1 is not None, "stage 1 produced no reference path; change the seed"
    assert np.array_equal(path1, opath1) and np.array_equal(act1, oact1)
    # stage 2: an obstacle appears on the reference path, re-plan from the start with the path as guide
    cells = np.floor(np.stack([10.0 - path1[:, 1], path1[:, 0] + 10.0], axis=1)).astype(int)
    k = int(len(cells) * 0.6)
    maze2 = maze.copy()
    maze2[cells[k][0], cells[k][1]] = 1
    planner.update_maze(maze2)
    planner.init_main_path = path1.copy()
    planner.time_budget = n2
    planner.reset(start_state=start, goal_state=goal)
    path2, act2 = _plan_with_candidate_clock(planner, counter)
    par2, st2 = _tree_of(planner)
    pl2 = ORRT.OraclePlanner(maze2, start, goal, ActionTape(seed).sampler(), run_type=1, init_main_path=path1)
    r2, opath2, oact2 = pl2.plan(ORRT.RandomTape(42), n2, batch=1)
    assert not pl2.sticky_triggered
    assert np.array_equal(np.array(pl2.tree.parents, dtype=np.int32), par2), "run_type 1 stage 2 parents"
    assert np.array_equal(np.array(pl2.tree.states), st2)
    assert (path2 is None) == (opath2 is None)
    if path2 is not None:
        assert np.array_equal(path2, opath2) and np.array_equal(act2, oact2)
    out.update(rt1_seed=np.array(seed), rt1_budgets=np.array([n1, n2]), rt1_start=start, rt1_goal=goal,
               rt1_parents1=par1, rt1_states1=st1, rt1_path1=path1, rt1_actions1=act1,
               rt1_maze2=maze2, rt1_parents2=par2, rt1_states2=st2,
               rt1_path2=np.zeros((0, 6), np.float32) if path2 is None else path2,
               rt1_actions2=np.zeros((0, 2), np.float32) if act2 is None else act2,
               rt1_has_path2=np.array(path2 is not None))
    print(f"run_type 1: stage 1 {len(par1)} nodes, stage 2 {len(par2)} nodes, path2={'yes' if path2 is not None else 'none'}; "
          f"oracle planner == reference planner; obstacle-ahead {ahead.mean():.2f} true on {len(poses)} poses")


def gen_trace_schedule(out):
    """planners/RRT.py:149-152 with a three-entry prop_duration: the edge length of a visit follows the parent's visit
    count.  Reference planner (B = 1) on boxes; the oracle planner with the same schedule must build the same tree."""
    scn = ("boxes", "boxes", 17, 2, 45, 2, 17)
    sched = (128, 64, 32)
    n = 400
    ref = run_reference_planner(scn, n, tape_seed=29, prop_duration=sched)
    pl = ORRT.OraclePlanner(ref["maze"], ref["start"], ref["goal"], ActionTape(29).sampler(), prop_duration=list(sched))
    reached, path, actions = pl.plan(ORRT.RandomTape(42), n, batch=1)
    assert not pl.sticky_triggered
    assert np.array_equal(np.array(pl.tree.parents, dtype=np.int32), ref["parents"]), "schedule trace: parents"
    assert np.array_equal(np.array(pl.tree.states), ref["states"])
    assert reached == ref["reached"] and pl.iterations == ref["iterations"]
    assert (path is None) == (ref["path"] is None)
    if path is not None:
        assert np.array_equal(path, ref["path"]) and np.array_equal(actions, ref["actions"])
    lens = np.array([0 if e is None else len(e) for e in pl.tree.edge_actions])
    assert len(set(lens[1:].tolist())) > 1, "the schedule never changed the edge length: pick another seed"
    out.update(sched_schedule=np.array(sched), sched_seed=np.array(29), sched_budget=np.array(n), sched_start=ref["start"],
               sched_goal=ref["goal"], sched_parents=ref["parents"], sched_states=ref["states"], sched_reached=np.array(ref["reached"]),
               sched_iterations=np.array(ref["iterations"]), sched_edge_actions=lens,
               sched_path=np.zeros((0, 6), np.float32) if ref["path"] is None else ref["path"],
               sched_actions=np.zeros((0, 2), np.float32) if ref["actions"] is None else ref["actions"])
    print(f"prop_duration {sched}: {len(ref['parents'])} nodes, edge lengths {sorted(set(lens[1:].tolist()))}, "
          f"reached {ref['reached']}; oracle planner == reference planner")


def gen_traces(out):
    scns = {
        "race": ("Race Track", "Race_Track", 1, 1, 270, 1, 10),
        "boxes": ("boxes", "boxes", 17, 2, 45, 2, 17),
        "rlarge2": ("random large2", "random_large", 1, 3, 0, 7, 10),
        "easy": ("easy (build-defined)", "val_maze_7", 1, 1, 0, 1, 4),
    }
    budgets = {"race": 400, "boxes": 400, "rlarge2": 400, "easy": 600}
    for tag, scn in scns.items():
        seed = 2026
        r = run_reference_planner(scn, budgets[tag], seed)
        # the oracle planner on the same tapes, B = 1
        pl = ORRT.OraclePlanner(r["maze"], r["start"], r["goal"], ActionTape(seed).sampler())
        reached, path, actions = pl.plan(ORRT.RandomTape(42), budgets[tag], batch=1)
        assert not pl.sticky_triggered, "trace hit the sticky-done defect; pick another seed"
        t = pl.tree
        assert reached == r["reached"], (tag, reached, r["reached"])
        assert np.array_equal(np.array(t.parents, dtype=np.int32), r["parents"]), tag
        assert np.array_equal(np.array(t.states), r["states"]), tag
        assert pl.iterations == r["iterations"], (pl.iterations, r["iterations"])
        assert np.array_equal(path, r["path"]) and np.array_equal(actions, r["actions"]), tag
        out[f"trace_{tag}_scenario"] = np.array(scn[2:], dtype=np.int64)
        out[f"trace_{tag}_maze_name"] = np.array(scn[1])
        out[f"trace_{tag}_budget"] = np.array(budgets[tag])
        out[f"trace_{tag}_tape_seed"] = np.array(seed)
        out[f"trace_{tag}_parents"] = r["parents"]
        out[f"trace_{tag}_states"] = r["states"]
        out[f"trace_{tag}_reached"] = np.array(r["reached"])
        out[f"trace_{tag}_path"] = r["path"]
        out[f"trace_{tag}_actions"] = r["actions"]
        out[f"trace_{tag}_iterations"] = np.array(r["iterations"])
        print(f"trace {tag}: {len(r['parents'])} nodes, reached={r['reached']}, "
              f"{r['iterations']} chunk-iterations, oracle planner == reference planner")


def gen_prob_maps(out):
    """run_type >= 2 sampling-probability maps: prob_sampling_utils.gaussian_map / combine_log_blend and the EDT
    prior of car_env.py:100-101, evaluated by the reference's own functions."""
    import prob_sampling_utils as ref_psu
    from scipy.ndimage import distance_transform_edt
    from oracle import prob_maps as PM
    maze = load_maze("boxes")
    rng = np.random.default_rng(23)
    pairs = np.floor(rng.uniform(1, 19, size=(24, 4)))
    pairs[0, 2:] = pairs[0, :2]                       # robot == goal
    pairs[1] = [3.0, 3.0, 18.0, 15.0]                 # the reference's own demo (prob_sampling_utils.py:181-183)
    gauss, blend = [], []
    mazes = [maze]
    for k in range(3):
        m = maze.copy()
        rc = rng.integers(2, 18, size=(4, 2))
        m[rc[:, 0], rc[:, 1]] = 1
        mazes.append(m)
    priors = []
    for m in mazes:
        pr = distance_transform_edt(1 - m)
        pr = pr / np.sum(pr)
        assert np.array_equal(pr, PM.edt_prior(m))
        priors.append(pr)
    for i, (rx, ry, gx, gy) in enumerate(pairs):
        pdf, mean, sig = ref_psu.gaussian_map((rx, ry), (gx, gy))
        opdf, omean, osig = PM.gaussian_map((rx, ry), (gx, gy))
        assert np.array_equal(pdf, opdf) and np.array_equal(mean, omean) and np.array_equal(sig, osig), i
        post = ref_psu.combine_log_blend(priors[i % len(priors)], pdf)
        assert np.array_equal(post, PM.combine_log_blend(priors[i % len(priors)], opdf)), i
        gauss.append(pdf)
        blend.append(post)
    out["probmap_pairs"] = pairs
    out["probmap_mazes"] = np.array(mazes)
    out["probmap_priors"] = np.array(priors)
    out["probmap_gauss"] = np.array(gauss)
    out["probmap_blend"] = np.array(blend)
    print(f"probability maps: {len(pairs)} gaussian / log-blend maps on {len(mazes)} priors, oracle == reference")


def gen_traces_run_type23(out):
    """run_type 2 (EDT-prior cell sampling) and 3 (prior x start->goal Gaussian) planner traces: the reference
    planner on the CarEnv restatement (base_planner.py:157-160,181-184, RRT.py:124-125)."""
    maze = load_maze("boxes")
    env0 = ORRT.OracleCarEnv(maze_map=maze)
    start = np.array([*env0.cell_rowcol_to_xy(np.array([17, 2])), np.deg2rad(45.0), 0.0, 0.0, 0.0])
    goal = np.array([*env0.cell_rowcol_to_xy(np.array([2, 17])), 0.0, 0.0, 0.0, 0.0])
    for rt, seed, n in ((2, 31, 250), (3, 32, 250)):
        counter = {"cand": 0}
        planner, env = _ref_planner(maze.copy(), start, goal, seed, n, rt, counter)
        planner.reset()
        path, act = _plan_with_candidate_clock(planner, counter)
        par, st = _tree_of(planner)
        pl = ORRT.OraclePlanner(maze, start, goal, ActionTape(seed).sampler(), run_type=rt)
        _, opath, oact = pl.plan(ORRT.RandomTape(42), n, batch=1)
        assert not pl.sticky_triggered
        assert np.array_equal(np.array(pl.tree.parents, dtype=np.int32), par), f"run_type {rt} parents"
        assert np.array_equal(np.array(pl.tree.states), st)
        assert (path is None) == (opath is None)
        if path is not None:
            assert np.array_equal(path, opath) and np.array_equal(act, oact)
        out[f"rt{rt}_seed"], out[f"rt{rt}_budget"] = np.array(seed), np.array(n)
        out[f"rt{rt}_parents"], out[f"rt{rt}_states"] = par, st
        out[f"rt{rt}_prob_map"] = pl.sampling_map()
        out[f"rt{rt}_path"] = np.zeros((0, 6), np.float32) if path is None else path
        out[f"rt{rt}_actions"] = np.zeros((0, 2), np.float32) if act is None else act
        print(f"run_type {rt}: {len(par)} nodes, path={'yes' if path is not None else 'none'}; oracle planner == reference planner")


def _driver_functions(names):
    """Functions of run_scenarios_with_lidar_DiTree.py taken from its text: the script's top-level imports
    (playsound, minari, gymnasium, drone_env ...) are absent here, so it cannot be imported as a module."""
    import ast
    path = os.path.join(REF, "run_scenarios_with_lidar_DiTree.py")
    tree = ast.parse(open(path).read(), filename=path)
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names], type_ignores=[])
    ns = {"np": np, "plt": None}
    exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def _drive(maze, start, waypoints_rc, v_des=0.6, max_steps=4000, switch=0.25):
    """Test scaffolding: a waypoint-following controller on the oracle dynamics that yields a drivable
    (states f32, actions f32) plan through a corridor maze, as a planner would return it."""
    wp = [np.array(G.cell_rowcol_to_xy(rc, maze)) for rc in waypoints_rc]
    s = np.asarray(start, dtype=np.float64).copy()
    acts, states, k = [], [s.copy()], 0
    for _ in range(max_steps):
        while k < len(wp) - 1 and np.hypot(*(wp[k] - s[:2])) < switch:
            k += 1
        if k == len(wp) - 1 and np.hypot(*(wp[k] - s[:2])) < 0.2:
            break
        err = np.arctan2(wp[k][1] - s[1], wp[k][0] - s[0]) - s[2]
        err = (err + np.pi) % (2 * np.pi) - np.pi
        a1 = np.clip((np.clip(0.8 * err, -0.35, 0.35) - s[5]) * 10, -2, 2)
        vd = v_des * (0.4 if abs(err) > 0.5 else 1.0)
        a0 = np.clip((np.clip(0.5 * (vd - s[3]) + 0.06, -0.5, 1.0) - s[4]) * 20, -10, 10)
        a = np.array([a0, a1], dtype=np.float32)
        s = G.car_step(s[None], a.astype(np.float64)[None])[0]
        assert not G.is_colliding_car(s[None, :3], maze)[0]
        acts.append(a)
        states.append(s.copy())
    return np.array(states, dtype=np.float32), np.array(acts, dtype=np.float32)


def gen_online(out):
    """Online driver steps (run_scenarios_with_lidar_DiTree.py:112-127,158-181,470-506): lidar scan of the true
    maze written into the known / scanned mazes, path-crossing check, and the action-execution loop, driven with the
    reference planner's propagate_action_sequence_env and the reference's Lidar2DSim."""
    from oracle import online as OO
    ref_scan, ref_check = _driver_functions(["scan_and_update_maze", "check_no_obstacles_in_path"])
    tr = np.load(os.path.join(HERE, "traces.npz"))
    cases = []

    def scenario(tag):
        maze = load_maze(str(tr[f"trace_{tag}_maze_name"]))
        sr, sc, sdeg, gr, gc = [int(v) for v in tr[f"trace_{tag}_scenario"]]
        start = np.array([*G.cell_rowcol_to_xy([sr, sc], maze), np.deg2rad(float(sdeg)), 0, 0, 0])
        goal = np.array([*G.cell_rowcol_to_xy([gr, gc], maze), 0, 0, 0, 0])
        return maze, start, goal, tr[f"trace_{tag}_path"], tr[f"trace_{tag}_actions"]

    def run_case(tag, scn, true_maze, acts, goal_state):
        maze, start, _, path, _ = scn
        counter = {"cand": 0}
        planner, env = _ref_planner(maze.copy(), start, goal_state, 1, 10, 1, counter)
        env.lidar2dsim = RefLidar()
        planner.reset(start_state=start, goal_state=goal_state)
        known, scanned = maze.copy(), maze.copy()
        oknown, oscanned = maze.copy(), maze.copy()
        ref_scan(planner, known, true_maze, scanned)                       # :415-416 initial scan (float32 env state)
        OO.scan_and_update_maze(env.state, oknown, true_maze, oscanned)
        assert np.array_equal(known, oknown) and np.array_equal(scanned, oscanned), "initial scan"
        res = {"known0": known.copy(), "scanned0": scanned.copy()}
        # :432-506 (run_type < 4) restated around the reference pieces
        env.reset_done()
        cur = start.copy()
        idx, obstacle, t_acc, done = 0, -1, 0, False
        executed = []
        env.set_state(cur)
        while idx < acts.shape[0] and obstacle < 0 and not done:
            nxt, done, _, visited = planner.propagate_action_sequence_env(cur, acts[idx, np.newaxis])
            if done is None:
                break
            executed.append(visited[0, 1, :])
            cur = nxt
            env.set_state(cur)
            idx += 1
            t_acc += env.dt
            if t_acc > env.lidar2dsim.scan_time:
                ref_scan(planner, known, true_maze, scanned)
                obstacle = ref_check(planner, scanned, path)
                t_acc = 0
        event = OO.EV_COLLISION if done is None else OO.EV_GOAL if done else OO.EV_OBSTACLE if obstacle >= 0 \
            else OO.EV_ACTIONS_DONE
        o = OO.follow_plan(start, acts, 0, path, oknown, true_maze, oscanned, env.goal, dt=env.dt,
                           scan_time=env.lidar2dsim.scan_time)
        assert o["event"] == event and o["action_idx"] == idx and o["obstacle_idx"] == obstacle, (tag, o["event"], event)
        assert np.array_equal(o["executed"], np.array(executed).reshape(-1, 6)) and np.array_equal(o["state"], cur)
        assert np.array_equal(known, oknown) and np.array_equal(scanned, oscanned), "mazes after the loop"
        res.update(maze=maze, start=start, path=path, true=true_maze, actions=acts,
                   goal_xy=np.asarray(env.goal, dtype=np.float64), known1=known.copy(), scanned1=scanned.copy(),
                   executed=np.array(executed).reshape(-1, 6), result=np.array([event, idx, obstacle], dtype=np.int64),
                   state=cur)
        note = f"{tag}: event {event} after {idx}/{len(acts)} actions, obstacle idx {obstacle}"
        return res, note, planner

    def keep(tag, res, note):
        for k, v in res.items():
            out[f"online_{tag}_{k}"] = v
        cases.append(note)

    boxes = scenario("boxes")
    maze, start, goal, path, actions = boxes
    H = maze.shape[0]
    cells = np.floor(np.stack([H / 2 - path[:, 1], path[:, 0] + maze.shape[1] / 2], axis=1)).astype(int)
    # visible: an obstacle early on the path is seen by the first scan
    tm = maze.copy()
    k = int(len(cells) * 0.25)
    tm[cells[k][0], cells[k][1]] = 1
    res, note, planner = run_case("visible", boxes, tm, actions, goal)
    assert res["result"][0] == OO.EV_OBSTACLE
    keep("visible", res, note)
    # free: the whole plan is executed
    res, note, _ = run_case("free", boxes, maze.copy(), actions, goal)
    assert res["result"][0] == OO.EV_ACTIONS_DONE
    keep("free", res, note)
    exe = res["executed"]
    # goal: the goal cell lies on the executed trajectory
    gs = goal.copy()
    gs[:2] = exe[len(exe) // 2, :2]
    res, note, _ = run_case("goal", boxes, maze.copy(), actions, gs)
    assert res["result"][0] == OO.EV_GOAL
    keep("goal", res, note)
    # collision: an open-loop swerve that leaves the planned corridor
    swerve = np.stack([np.full(600, 1.5), 0.6 * np.sign(np.sin(np.arange(600) / 40.0))], axis=1).astype(np.float32)
    res, note, _ = run_case("collision", boxes, maze.copy(), swerve, goal)
    assert res["result"][0] == OO.EV_COLLISION
    keep("collision", res, note)
    # hidden: in a corridor maze an obstacle behind a corner is found by a later scan
    # (the reference's Lidar2DSim only works on square maps: its clip bounds are swapped, lidar_2d_sim.py:89-91)
    rmaze = load_maze("val_maze_15")
    rstart = np.array([*G.cell_rowcol_to_xy([1, 1], rmaze), np.deg2rad(270.0), 0, 0, 0])
    rgoal = np.array([*G.cell_rowcol_to_xy([13, 13], rmaze), 0, 0, 0, 0])
    rpath, ract = _drive(rmaze, rstart, [[5, 1], [9, 1], [9, 3], [9, 5]])
    race = (rmaze, rstart, rgoal, rpath, ract)
    tm = rmaze.copy()
    tm[9, 4] = 1
    res, note, _ = run_case("hidden", race, tm, ract, rgoal)
    assert res["result"][0] == OO.EV_OBSTACLE and res["result"][1] > 100, note
    keep("hidden", res, note)
    res, note, _ = run_case("track", race, rmaze.copy(), ract, rgoal)
    assert res["result"][0] == OO.EV_ACTIONS_DONE
    keep("track", res, note)
    # path-crossing check alone on perturbed scanned mazes (float32 path arithmetic)
    rng = np.random.default_rng(11)
    expected, marks = [], []
    for _ in range(32):
        sc = maze.copy()
        rc = rng.integers(1, 19, size=(3, 2))
        sc[rc[:, 0], rc[:, 1]] = 1
        expected.append(ref_check(planner, sc, path))
        assert expected[-1] == OO.check_no_obstacles_in_path(sc, path)
        marks.append(rc)
    out["online_check_marks"], out["online_check_expected"] = np.array(marks), np.array(expected)
    print("online driver steps: " + "; ".join(cases) + f"; path check {np.sum(np.array(expected) >= 0)}/32 crossings; "
          "oracle == reference")


def gen_path_after_obstacle(out):
    """The reference's RRT_Planner.extract_path_after_obstacle (planners/RRT.py:83-111) itself, on float32 paths with float64
    AND float32 env states -- incl. states placed so that the nearest path point differs between float32 and float64
    arithmetic (the reference's result then depends on the dtype of env.state: numpy promotion is part of its behaviour)."""
    maze = load_maze("boxes")
    rng = np.random.default_rng(77)

    class _E:
        def __init__(self, st):
            self.state = st

        def cell_xy_to_rowcol(self, xy, floor_enable=True):
            return G.cell_xy_to_rowcol(xy, maze, floor_enable=floor_enable)
    cases = []
    for k in range(40):
        # an L-shaped path through the free bottom corridor (row 18) and up column 18, with jitter, as float32
        n1, n2 = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        a, b, c = G.cell_rowcol_to_xy([18, 1], maze), G.cell_rowcol_to_xy([18, 18], maze), G.cell_rowcol_to_xy([1, 18], maze)
        p = np.concatenate([a + (b - a) * np.linspace(0, 1, n1)[:, None], b + (c - b) * np.linspace(0, 1, n2)[1:, None]])
        p = (p + rng.normal(0, 0.02, p.shape)).astype(np.float32)
        path = np.zeros((len(p), 6), dtype=np.float32)
        path[:, :2] = p
        mz = maze.copy()
        if k % 4 != 3:                                   # an obstacle written onto the path (as a lidar scan would)
            i = int(rng.integers(10, len(p) - 5))
            rc = G.cell_xy_to_rowcol(p[i].astype(np.float64), maze)
            mz[int(rc[0]), int(rc[1])] = 1
        j = int(rng.integers(0, len(p) - 1))
        if k % 2 == 0:                                   # between two path points, a hair off the f32 bisector
            mid = (p[j].astype(np.float64) + p[j + 1].astype(np.float64)) / 2
            st = np.zeros(6)
            st[:2] = mid + rng.normal(0, 1e-9, 2)
        else:
            st = np.zeros(6)
            st[:2] = p[j].astype(np.float64) + rng.normal(0, 0.3, 2)
        for dt in (np.float64, np.float32):
            pl = ref_rrt.RRT_Planner.__new__(ref_rrt.RRT_Planner)
            pl.env, pl.maze, pl.init_main_path = _E(st.astype(dt)), mz, path
            exp = pl.extract_path_after_obstacle()
            mine = ORRT.path_after_obstacle(path, st.astype(dt), mz)
            assert exp.dtype == mine.dtype and np.array_equal(exp, mine), (k, dt)
            cases.append((path[:, :2].copy(), st[:2].copy(), dt == np.float32, mz, exp))
    out["pao_n"] = np.array(len(cases))
    differ = 0
    for i, (p, st, f32, mz, exp) in enumerate(cases):
        out[f"pao_{i}_path"], out[f"pao_{i}_state"], out[f"pao_{i}_f32"] = p, st, np.array(f32)
        out[f"pao_{i}_maze"] = np.packbits(mz.astype(np.uint8), axis=None)
        out[f"pao_{i}_expected"] = exp
        if i % 2 == 1 and len(cases[i - 1][4]) != len(exp):
            differ += 1
    print(f"extract_path_after_obstacle: {len(cases)} cases (float64 / float32 env states; {differ} pairs where the dtype changes the result), oracle == reference")


# --------------------------------------------------------------------------- ant (BASELINE config 3): glue pinned by the reference
def ant_poses(rng, maze, n, sg):
    """Poses for is_colliding_ant: uniform over the scaled map (+ a margin outside), a third snapped onto cell boundaries /
    exactly one ball radius (1.2) or the corner distance from them, quaternions upright, tilted past 90 deg, and random."""
    H, W = maze.shape
    s = np.zeros((n, 29))
    s[:, 0] = rng.uniform(-W / 2 * sg - 2, W / 2 * sg + 2, n)
    s[:, 1] = rng.uniform(-H / 2 * sg - 2, H / 2 * sg + 2, n)
    nb = n // 4
    off = [0.0, 1e-12, -1e-12, 1.2, -1.2, 1.2 + 1e-12, -1.2 - 1e-12, 1.2 - 1e-12, 0.8, -0.8, 2.0, -2.0]
    s[:nb, 0] = np.round(s[:nb, 0] / sg * 2) / 2 * sg + rng.choice(off, nb)
    s[nb:2 * nb, 1] = np.round(s[nb:2 * nb, 1] / sg * 2) / 2 * sg + rng.choice(off, nb)
    # near cell corners: the diagonal test (map_utils.py:202-215)
    k = slice(2 * nb, 3 * nb)
    ang = rng.uniform(0, 2 * np.pi, nb)
    rad = rng.choice([1.2, 1.2 - 1e-9, 1.2 + 1e-9, 0.9, 1.5], nb)
    s[k, 0] = np.round(s[k, 0] / sg) * sg + rad * np.cos(ang)
    s[k, 1] = np.round(s[k, 1] / sg) * sg + rad * np.sin(ang)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    up = rng.random(n) < 0.6
    q[up] = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.25, (int(up.sum()), 4))
    edge = rng.random(n) < 0.1                       # body z axis (almost) horizontal: 1 - 2 (qx^2 + qy^2) ~ 0
    th = rng.uniform(0, 2 * np.pi, n)
    q[edge, 0], q[edge, 3] = 0.0, np.sqrt(0.5)
    q[edge, 1] = np.sqrt(0.5) * np.cos(th[edge]) + rng.choice([0.0, 1e-9, -1e-9], int(edge.sum()))
    q[edge, 2] = np.sqrt(0.5) * np.sin(th[edge])
    s[:, 3:7] = q
    s[:, 2] = rng.uniform(0.3, 0.9, n)
    s[:, 7:] = rng.normal(0, 1, (n, 22))
    s[-3:, 0] = [np.nan, np.inf, -np.inf]            # NaN / inf positions: floor().astype(int) -> out of the map
    return s


def gen_ant_collision(out):
    """The reference's own is_colliding_ant(state, maze, 1.2, s_global) (common/map_utils.py:126-219, called at
    planners/base_planner.py:154-155) and goal test (:296-297) on 4096 states x 4 mazes; oracle == reference asserted."""
    from oracle import ant as OA
    rng = np.random.default_rng(401)
    sg = 4.0
    for name in ("Race_Track", "boxes", "random_huge", "narrow_short"):
        maze = load_maze(name)
        st = ant_poses(rng, maze, 4096, sg)
        with np.errstate(invalid="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                exp = np.array([bool(ref_mu.is_colliding_ant(s, maze, 1.2, sg)) for s in st])
        mine = OA.is_colliding_ant(st, maze, 1.2, sg)
        assert (exp == mine).all(), f"ant collision mismatch on {name}: {np.nonzero(exp != mine)[0][:10]}"
        out[f"antcol_{name}_states"] = st[:, :7].copy()
        out[f"antcol_{name}_expected"] = exp
        print(f"ant collision {name}: {exp.mean():.3f} colliding ({int((OA.body_z_up(st[:, 3:7]) < 0).sum())} upside down), oracle == reference")
    # a unit-scale call (ball_radius 0.1 defaults are the point-maze's; the engine takes both as parameters)
    maze = load_maze("boxes")
    st = ant_poses(rng, maze, 1024, 1.0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.array([bool(ref_mu.is_colliding_ant(s, maze, 0.3, 1.0)) for s in st])
    assert (exp == OA.is_colliding_ant(st, maze, 0.3, 1.0)).all()
    out["antcol_unit_states"], out["antcol_unit_expected"] = st[:, :7].copy(), exp


class _AntMazeData:
    """What planners/base_planner.py:81-92 reads from ``env.maze_data`` (gymnasium-robotics ``Maze``: map_length = rows,
    map_width = cols, centres already multiplied by maze_size_scaling)."""

    def __init__(self, maze, s):
        self.maze_map = np.asarray(maze)
        self.maze_size_scaling = s
        self.map_length, self.map_width = self.maze_map.shape
        self.x_map_center = self.map_width / 2 * s
        self.y_map_center = self.map_length / 2 * s

    def cell_xy_to_rowcol(self, xy):
        i = np.floor((self.y_map_center - xy[1]) / self.maze_size_scaling)
        j = np.floor((xy[0] + self.x_map_center) / self.maze_size_scaling)
        return np.array([i, j])

    def cell_rowcol_to_xy(self, rc):
        return np.array([(rc[1] + 0.5) * self.maze_size_scaling - self.x_map_center,
                         self.y_map_center - (rc[0] + 0.5) * self.maze_size_scaling])


class _AntEnvStandIn:
    """The surface of the gym AntMaze env the reference planner touches (planners/base_planner.py:81-92,278-298;
    planners/RRT.py:30,122 also read ``prob_map`` / set ``run_type``), with the MuJoCo step replaced by ``step_fn`` -- the
    tape or the build's surrogate model (oracle/ant.py).  Generation-time scaffolding: the REFERENCE's planner code runs on it."""

    def __init__(self, maze, s_global, desired_xy, step_fn, counter):
        self.maze_data = _AntMazeData(maze, s_global)
        self.prob_map = np.zeros_like(np.asarray(maze, dtype=np.float64))
        self.ant_env = self
        self.desired = np.asarray(desired_xy, dtype=np.float64)
        self.step_fn, self.counter = step_fn, counter
        self.state = np.zeros(29)
        self.chunk, self.i, self.last_cand = -1, 0, -1

    def reset(self, options=None, **kw):
        return None, {}

    def set_state(self, qpos, qvel):                        # base_planner.py:278-279: once per chunk
        self.state = np.concatenate([qpos, qvel]).astype(np.float64)
        cand = self.counter["cand"] - 1
        if cand != self.last_cand:
            self.chunk, self.last_cand = -1, cand
        self.chunk += 1
        self.i = 0

    def step(self, action):
        cand = self.counter["cand"] - 1
        self.state = self.step_fn(np.array([cand]), self.chunk, self.i, self.state[None], np.asarray(action, dtype=np.float64)[None])[0]
        self.i += 1
        obs = {"achieved_goal": self.state[:2].copy(), "desired_goal": self.desired.copy(), "observation": self.state[2:].copy()}
        return obs, 0.0, False, False, {}


class _AntRecordingSampler(torch.nn.Module):
    """Action tape + a record of what every sampler call was handed (RRT.py:146-147,168-171,186-190)."""

    def __init__(self, tape, counter):
        super().__init__()
        self.tape, self.counter = tape, counter
        self.chunk, self.last_cand = 0, -1
        self.calls = []

    def forward(self, prev_states, prev_actions=None, goal=None, local_map=None):
        cand = self.counter["cand"] - 1
        if cand != self.last_cand:
            self.chunk, self.last_cand = 0, cand
        ps = np.asarray(prev_states)
        assert ps.ndim == 3 and ps.shape[0] == 1
        h = ps[0, -3:]
        hist = np.zeros((3, 29))
        hist[3 - len(h):] = h
        pa = np.zeros(8) if prev_actions is None else np.asarray(prev_actions)[-1]
        self.calls.append(dict(cand=cand, chunk=self.chunk, hist=hist, n_hist=len(h), prev=pa, has_prev=prev_actions is not None and len(prev_actions) > 0,
                               goal=np.asarray(goal, dtype=np.float64).copy(), lmap=np.asarray(local_map.cpu().numpy() if hasattr(local_map, "cpu") else local_map)[0].astype(np.uint8)))
        a = self.tape.actions(np.array([cand]), self.chunk)
        self.chunk += 1
        return a


def run_reference_ant_planner(maze_name, start_rc, goal_rc, n_candidates, seed, dynamics):
    """The reference's RRT_Planner(env_id='antmaze') (planners/RRT.py:113-257 with base_planner.py's ant branches) on the
    stand-in env; cfgs/antmaze.yaml + run_scenarios.py:123-132: action_horizon 2, prop_duration [48], pred_horizon 16,
    local map 16 @ 0.8, s_global 4."""
    from oracle import ant as OA
    import random
    maze = load_maze(maze_name)
    sg, A, H, P = 4.0, 2, 48, 16
    md = _AntMazeData(maze, sg)
    start = np.zeros(29)
    start[:2] = md.cell_rowcol_to_xy(np.array(start_rc))
    start[2], start[3] = 0.75, 1.0                                   # run_scenarios.py:228-230
    if dynamics == "model":
        start[7:15] = np.tile([0.0, OA.AntModel.ank_rest], 4)
    goal = np.zeros(29)
    goal[:2] = md.cell_rowcol_to_xy(np.array(goal_rc))
    desired = goal[:2] + np.array([0.3, -0.2])                       # the env's goal carries position noise
    atape = OA.AntActionTape(seed, P)
    if dynamics == "tape":
        otape = OA.AntObsTape(seed + 1, maze, sg, H // A, A, desired_xy=desired, goal_every=29, step=0.12)
        step_fn = otape.step_fn()
    else:
        step_fn = lambda cand, chunk, i, cur, act: OA.ant_model_step(cur, act)     # noqa: E731
    counter = {"cand": 0}
    env = _AntEnvStandIn(maze, sg, desired, step_fn, counter)
    smp = _AntRecordingSampler(atape, counter)
    random.seed(42)
    np.random.seed(42)
    planner = ref_rrt.RRT_Planner(start, goal, env_id="antmaze", environment=env, sampler=smp, prediction_type="actions",
                                  action_horizon=A, local_map_size=16, local_map_scale=0.8, global_map_scale=sg,
                                  goal_conditioning_bias=0.85, prop_duration=[H], time_budget=n_candidates, max_iter=300,
                                  verbose=False)
    planner.device = "cpu"
    orig_nearest = planner.nearest_node

    def counted_nearest(sample):
        counter["cand"] += 1
        return orig_nearest(sample)
    planner.nearest_node = counted_nearest
    real_time = ref_rrt.time.time
    ref_rrt.time.time = lambda: float(counter["cand"])
    try:
        random.seed(42)
        np.random.seed(42)
        path, actions = planner.plan()
    finally:
        ref_rrt.time.time = real_time
    parents, states = _tree_of(planner)
    # the oracle planner, B = 1, same tapes
    pl = OA.OracleAntPlanner(maze, start, goal, desired, atape.sampler(), step_fn, edge_length=H, action_horizon=A)
    rec = []
    inner = pl.sampler

    def rec_sampler(cand_idx, chunk, hist, prev_a, has_prev, cond_goal, lm):
        for k, c in enumerate(cand_idx):
            rec.append((int(c), chunk, hist[k], prev_a[k], bool(has_prev[k]), cond_goal[k], lm[k]))
        return inner(cand_idx, chunk, hist, prev_a, has_prev, cond_goal, lm)
    pl.sampler = rec_sampler
    reached, opath, oactions = pl.plan(ORRT.RandomTape(42), counter["cand"], batch=1)
    assert np.array_equal(parents, np.array(pl.parents)), "ant trace: parents differ"
    assert np.array_equal(states, np.array(pl.states)), "ant trace: node states differ"
    assert planner.results["iterations"] == pl.iterations, (planner.results["iterations"], pl.iterations)
    assert (path is None) == (opath is None) and (path is None or (np.array_equal(path, opath) and np.array_equal(actions, oactions)))
    assert len(rec) == len(smp.calls)
    for (c, j, hist, pa, hp, g, lm), call in zip(rec, smp.calls):
        assert (c, j) == (call["cand"], call["chunk"]) and len(hist) == call["n_hist"] and hp == call["has_prev"]
        assert np.array_equal(hist, call["hist"][3 - call["n_hist"]:]) and np.array_equal(g, call["goal"])
        assert (not hp) or np.array_equal(pa, call["prev"])
        assert np.array_equal(lm.astype(np.uint8), call["lmap"])
    goal_idx = None if not reached else pl.goal_node
    return dict(maze_name=maze_name, start=start, goal=goal, desired=desired, seed=seed, parents=parents, states=states,
                reached=bool(reached), path=path, actions=actions, iterations=planner.results["iterations"],
                candidates=counter["cand"], calls=smp.calls, goal_node=-1 if goal_idx is None else goal_idx)


def gen_ant_traces(out):
    cases = [("tape_boxes", "boxes", (17, 2), (2, 17), 260, 11, "tape"),
             ("tape_xlarge", "random_xlarge", (1, 1), (3, 7), 260, 12, "tape"),
             ("tape_val7", "val_maze_7", (1, 5), (5, 3), 200, 13, "tape"),
             ("model_boxes", "boxes", (17, 2), (2, 17), 160, 14, "model"),
             ("model_val7", "val_maze_7", (1, 5), (5, 3), 160, 15, "model")]
    for tag, maze_name, s_rc, g_rc, n, seed, dyn in cases:
        r = run_reference_ant_planner(maze_name, s_rc, g_rc, n, seed, dyn)
        pre = f"anttrace_{tag}_"
        for k in ("start", "goal", "desired", "parents", "states"):
            out[pre + k] = r[k]
        out[pre + "maze"] = load_maze(maze_name)
        out[pre + "meta"] = np.array([r["seed"], r["candidates"], r["iterations"], int(r["reached"]), r["goal_node"], 1 if dyn == "model" else 0])
        out[pre + "path"] = np.zeros((0, 29), np.float32) if r["path"] is None else r["path"]
        out[pre + "actions"] = np.zeros((0, 8), np.float32) if r["actions"] is None else r["actions"]
        calls = r["calls"][:1500]                       # what the first sampler calls were handed
        out[pre + "call_key"] = np.array([[c["cand"], c["chunk"], c["n_hist"], int(c["has_prev"])] for c in calls], dtype=np.int32)
        out[pre + "call_hist"] = np.array([c["hist"] for c in calls])
        out[pre + "call_prev"] = np.array([c["prev"] for c in calls])
        out[pre + "call_goal"] = np.array([c["goal"] for c in calls])
        out[pre + "call_lmap"] = np.packbits(np.array([c["lmap"] for c in calls]), axis=None)
        print(f"ant trace {tag}: {r['candidates']} candidates, {len(r['parents'])} nodes, {r['iterations']} chunk iterations, "
              f"reached {r['reached']}, {len(r['calls'])} sampler calls -- oracle == reference planner")


def main():
    if sys.argv[1:] == ["probmaps"]:          # quick check of the run_type >= 2 pieces only (writes nothing)
        gen_prob_maps({})
        gen_traces_run_type23({})
        return
    if sys.argv[1:] == ["ant"]:               # add the ant sampler fixtures to network.npz (other entries kept)
        net = dict(np.load(os.path.join(HERE, "network.npz"), allow_pickle=False))
        gen_sampler_ant(net)
        np.savez_compressed(os.path.join(HERE, "network.npz"), **net)
        return
    if sys.argv[1:] == ["pao"]:               # extract_path_after_obstacle cases -> traces.npz (other entries kept)
        tr = dict(np.load(os.path.join(HERE, "traces.npz"), allow_pickle=False))
        gen_path_after_obstacle(tr)
        np.savez_compressed(os.path.join(HERE, "traces.npz"), **tr)
        return
    if sys.argv[1:] == ["antglue"]:           # ant collision / goal / planner-trace fixtures -> ant.npz
        ant = {}
        gen_ant_collision(ant)
        gen_ant_traces(ant)
        np.savez_compressed(os.path.join(HERE, "ant.npz"), **ant)
        return
    if sys.argv[1:] == ["dyntext"]:           # add the source-text dynamics vectors to geometry.npz (other entries kept)
        geo = dict(np.load(os.path.join(HERE, "geometry.npz"), allow_pickle=False))
        gen_dynamics_from_text(geo)
        np.savez_compressed(os.path.join(HERE, "geometry.npz"), **geo)
        return
    if sys.argv[1:] == ["schedule"]:          # add the prop_duration-schedule trace to traces.npz (other entries kept)
        tr = dict(np.load(os.path.join(HERE, "traces.npz"), allow_pickle=False))
        gen_trace_schedule(tr)
        np.savez_compressed(os.path.join(HERE, "traces.npz"), **tr)
        return
    if sys.argv[1:] == ["online"]:
        online = {}
        gen_online(online)
        np.savez_compressed(os.path.join(HERE, "online.npz"), **online)
        return
    geo, net, traces, meta = {}, {}, {}, {}
    gen_collision(geo)
    gen_local_map(geo)
    gen_lidar(geo)
    gen_kdtree(geo)
    gen_dynamics(geo)
    gen_dynamics_from_text(geo)
    gen_sampler(net)
    gen_sampler_ant(net)
    gen_timesteps(meta)
    gen_unet(net)
    gen_traces(traces)
    gen_traces_run_type1(traces)
    gen_traces_run_type23(traces)
    gen_trace_schedule(traces)
    gen_path_after_obstacle(traces)
    gen_prob_maps(geo)
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **geo)
    np.savez_compressed(os.path.join(HERE, "network.npz"), **net)
    np.savez_compressed(os.path.join(HERE, "traces.npz"), **traces)
    online = {}
    gen_online(online)
    np.savez_compressed(os.path.join(HERE, "online.npz"), **online)
    ant = {}
    gen_ant_collision(ant)
    gen_ant_traces(ant)
    np.savez_compressed(os.path.join(HERE, "ant.npz"), **ant)
    with open(os.path.join(HERE, "timesteps.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote golden fixtures to", HERE)


if __name__ == "__main__":
    main()
