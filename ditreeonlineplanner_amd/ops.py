"""Torch-tensor front end of the C-ABI (one ``Context`` per process and device).

Tensors are containers only: every method passes ``data_ptr()`` of caller-owned CUDA
tensors plus sizes to libditree_hip.so and enqueues on torch's current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Round, RoundParams, Tree, check, lib

LIDAR_RAYS = 181

# metadata/carmaze.pt (reference) -- Observations_mean/std, Actions_mean/std; see DESIGN.md
CAR_NORM = np.array([0.0, 0.0, 0.0, 5.0, 0.5, 0.0,
                     5.0, 5.0, 3.141592653589793, 5.0, 0.5, 0.4,
                     0.45102226669605805, 0.0,
                     1.0061299587120194, 0.9234329966426903], dtype=np.float64)


def local_axis(n: int, scale: float) -> np.ndarray:
    """The reference's np.linspace(-L/2 + s/2, L/2 - s/2, n) (common/map_utils.py:422-423)."""
    L = n * scale
    return np.linspace(-L / 2 + scale / 2, L / 2 - scale / 2, n)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, name, dev):
    if not isinstance(t, torch.Tensor) or t.dtype != dtype or not t.is_contiguous() or t.device != dev:
        raise TypeError(f"{name}: need a contiguous {dtype} tensor on {dev}")


def _dbl(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _flt(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Context:
    """Owns a ``ditree_ctx`` on one GPU."""

    def __init__(self, device: int | torch.device | None = None):
        if not torch.cuda.is_available():
            raise _lib.DitreeLibraryError("no GPU visible: the expansion engine has no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index or 0)
        self._h = C.c_void_p()
        rc = lib().ditree_ctx_create(self.device.index, C.byref(self._h))
        if rc != 0:
            raise _lib.DitreeError(f"ditree_ctx_create failed ({rc})")
        self.maze_shape = None
        # One ctx holds ONE device maze and ONE weight set, but several planners / nets may share it (the facades'
        # default_context): whoever uploaded last is recorded here, and every user re-uploads when it is not the owner.
        self.maze_owner = None
        self.weights_owner = None

    def close(self):
        if getattr(self, "_h", None):
            lib().ditree_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ maze
    def upload_maze(self, maze, owner=None):
        """Copy the known maze to the device.  ``owner``: the object whose maze this is (``maze_owner`` afterwards);
        anonymous uploads (owner None) make every engine re-upload its own maze before its next launch."""
        m = np.ascontiguousarray(np.asarray(maze, dtype=np.float32))
        if m.ndim != 2:
            raise ValueError("maze must be 2-D")
        check(self._h, lib().ditree_upload_maze(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), m.shape[0],
                                                 m.shape[1], self.stream), "upload_maze")
        self.maze_shape = m.shape
        self.maze_owner = owner

    # ------------------------------------------------------------------ nearest node
    def nn_argmin(self, queries, node_xy, n_nodes=None, gather=None):
        """queries (B, >=2) f64, node_xy (N, 2) f64 -> idx (B,) i32.  ``gather`` =
        (node_state, node_last_action, node_has_prev) also returns the gathered rows."""
        dev = self.device
        _chk(queries, torch.float64, "queries", dev)
        _chk(node_xy, torch.float64, "node_xy", dev)
        B = queries.shape[0]
        N = node_xy.shape[0] if n_nodes is None else int(n_nodes)
        idx = torch.empty(B, dtype=torch.int32, device=dev)
        if gather is None:
            args = [None] * 6
            outs = None
        else:
            ns, na, nh = gather
            _chk(ns, torch.float64, "node_state", dev)
            _chk(na, torch.float64, "node_last_action", dev)
            _chk(nh, torch.uint8, "node_has_prev", dev)
            outs = (torch.empty(B, 6, dtype=torch.float64, device=dev),
                    torch.empty(B, 2, dtype=torch.float64, device=dev),
                    torch.empty(B, dtype=torch.uint8, device=dev))
            args = [_ptr(ns), _ptr(na), _ptr(nh), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2])]
        check(self._h, lib().ditree_nn_argmin(self._h, _ptr(queries), queries.shape[1], B, _ptr(node_xy), N,
                                               _ptr(idx), *args, self.stream), "nn_argmin")
        return idx if outs is None else (idx, *outs)

    # ------------------------------------------------------------------ local map
    def local_map(self, state, n=20, scale=0.2, s_global=1.0, scaled=False, active=None, out=None):
        """state (B, >= 3) f64: columns 0, 1, 2 are x, y and the rotation the reference passes (RRT.py:158-166)."""
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        if state.dim() != 2 or state.shape[1] < 3:
            raise ValueError("state must be (B, >= 3)")
        B = state.shape[0]
        if out is None:
            out = torch.empty(B, n, n, dtype=torch.float32, device=dev)
        ax, axp = _dbl(local_axis(n, scale))
        check(self._h, lib().ditree_local_map(self._h, _ptr(state), int(state.shape[1]), _ptr(active), B, n, axp,
                                               float(s_global), int(bool(scaled)), _ptr(out), self.stream), "local_map")
        return out

    # ------------------------------------------------------------------ cond vector (ant)
    def cond_vector_ant(self, obs, prev_action, has_prev, cond_goal, local_map_size, norm):
        """obs (B, n_hist <= 3, 29) f64, prev_action (B, 8) f64, has_prev (B,) u8, cond_goal (B, 2) f64 -> (B, 97) f32
        (policies/fm_policy.py:60-143, antmaze branch); norm = obs_mean[27] + obs_std[27] + act_mean[8] + act_std[8]."""
        dev = self.device
        _chk(obs, torch.float64, "obs", dev)
        _chk(prev_action, torch.float64, "prev_action", dev)
        _chk(has_prev, torch.uint8, "has_prev", dev)
        _chk(cond_goal, torch.float64, "cond_goal", dev)
        if obs.dim() != 3 or obs.shape[2] != 29 or not 1 <= obs.shape[1] <= 3 or prev_action.shape[1] != 8:
            raise ValueError("obs must be (B, 1..3, 29), prev_action (B, 8)")
        B = obs.shape[0]
        out = torch.empty(B, 97, dtype=torch.float32, device=dev)
        nm, nmp = _dbl(norm)
        if nm.size != 70:
            raise ValueError("norm: 27 + 27 + 8 + 8 doubles")
        check(self._h, lib().ditree_cond_vector_ant(self._h, _ptr(obs), int(obs.shape[1]), _ptr(prev_action), _ptr(has_prev),
                                                     _ptr(cond_goal), B, nmp, float(local_map_size), _ptr(out),
                                                     self.stream), "cond_vector_ant")
        return out

    # ------------------------------------------------------------------ cond vector
    def cond_vector(self, state, prev_action, has_prev, cond_goal, local_map_size=20, norm=CAR_NORM):
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        _chk(prev_action, torch.float64, "prev_action", dev)
        _chk(has_prev, torch.uint8, "has_prev", dev)
        _chk(cond_goal, torch.float64, "cond_goal", dev)
        B = state.shape[0]
        out = torch.empty(B, 7, dtype=torch.float32, device=dev)
        nm, nmp = _dbl(norm)
        check(self._h, lib().ditree_cond_vector(self._h, _ptr(state), _ptr(prev_action), _ptr(has_prev),
                                                 _ptr(cond_goal), B, nmp, float(local_map_size), _ptr(out),
                                                 self.stream), "cond_vector")
        return out

    # ------------------------------------------------------------------ rollout
    def car_rollout(self, state, actions, goal_xy, A=8, status=None, prev_action=None, has_prev=None, out=None, layout=None,
                    want_actions=True):
        """state (B,6) f64 [updated in place], actions (B, n>=A, 2) f64.
        Returns (status, states (B,A+1,6), actions_out (B,A,2), steps).  ``out``: a previous return value whose
        buffers are reused (the kernel writes every row, so no clearing is needed).
        ``layout``: "rows" = rows packed per candidate (the reference's arrays), "soa" = step-major / component-major /
        candidate-minor storage (every store of a wavefront is one contiguous 512-byte run; the returned tensors are
        permuted VIEWS of it with the same (B, A+1, 6) / (B, A, 2) shapes).  Default: "soa" from 4096 candidates up.
        ``want_actions=False``: the (A, 2) copy of the executed actions (the reference returns it, base_planner.py:318-320) is
        not written -- consumers that hold the actions already (an MPPI-style sweep) save 16 A bytes per rollout."""
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        _chk(actions, torch.float64, "actions", dev)
        B = state.shape[0]
        if layout is None:
            layout = "soa" if B >= 4096 else "rows"
        if layout not in ("rows", "soa"):
            raise ValueError("layout must be 'rows' or 'soa'")
        if status is None:
            status = torch.zeros(B, dtype=torch.int32, device=dev)
        if out is not None and tuple(out[1].shape) == (B, A + 1, 6) and (out[1].stride(0) == 1) == (layout == "soa"):
            _, states, aout, steps = out
        elif layout == "soa":
            states = torch.zeros(A + 1, 6, B, dtype=torch.float64, device=dev).permute(2, 0, 1)
            aout = torch.zeros(A, 2, B, dtype=torch.float64, device=dev).permute(2, 0, 1)
            steps = torch.zeros(B, dtype=torch.int32, device=dev)
        else:
            states = torch.zeros(B, A + 1, 6, dtype=torch.float64, device=dev)
            aout = torch.zeros(B, A, 2, dtype=torch.float64, device=dev)
            steps = torch.zeros(B, dtype=torch.int32, device=dev)
        g, gp = _dbl(goal_xy)
        if want_actions and aout is None:
            aout = (torch.zeros(A, 2, B, dtype=torch.float64, device=dev).permute(2, 0, 1) if layout == "soa"
                    else torch.zeros(B, A, 2, dtype=torch.float64, device=dev))
        sl = _lib.Strides(*states.stride())
        al = _lib.Strides(*aout.stride()) if aout is not None else _lib.Strides(2 * A, 2, 1)
        check(self._h, lib().ditree_car_rollout_ld(self._h, _ptr(state), _ptr(actions), actions.shape[1] * 2, _ptr(status), B, A,
                                                    gp, _ptr(states), C.byref(sl), _ptr(aout) if want_actions else None, C.byref(al),
                                                    _ptr(steps), _ptr(prev_action), _ptr(has_prev), self.stream), "car_rollout")
        return status, states, (aout if want_actions else None), steps

    # ------------------------------------------------------------------ ant: collision glue + the higher-DoF rollout slot
    def ant_collision(self, state, ball_radius=1.2, s_global=4.0):
        """is_colliding_ant(state, maze, ball_radius, s_global) (common/map_utils.py:126-219 as planners/base_planner.py:154-155
        calls it) for (B, >= 7) f64 states on the uploaded maze -> (B,) u8."""
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        if state.dim() != 2 or state.shape[1] < 7:
            raise ValueError("state must be (B, >= 7): x, y, z, qw, qx, qy, qz, ...")
        B = state.shape[0]
        out = torch.empty(B, dtype=torch.uint8, device=dev)
        check(self._h, lib().ditree_ant_collision(self._h, _ptr(state), int(state.shape[1]), B, float(ball_radius), float(s_global),
                                                   _ptr(out), self.stream), "ant_collision")
        return out

    def ant_rollout(self, state, actions, desired_goal, A=2, model=None, next_obs_tape=None, status=None, s_global=4.0,
                    ball_radius=1.2, goal_factor=0.45, layout=None, want_rows=True):
        """planners/base_planner.py:257-320 for B ant candidates (include/ditree.h ditree_ant_rollout): state (B, 29) f64
        [updated in place], actions (B, n >= A, 8) f64.  The env step is ``model`` (an ``_lib.AntModel``: the build's stand-in,
        NOT MuJoCo; True = the default constants) or row i of ``next_obs_tape`` (B, A, 29).
        -> (status, states (B, A+1, 29), actions_out (B, A, 8), steps); layout as car_rollout ("soa" = permuted views)."""
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        _chk(actions, torch.float64, "actions", dev)
        B = state.shape[0]
        if tuple(state.shape) != (B, 29) or actions.dim() != 3 or actions.shape[2] != 8 or actions.shape[1] < A:
            raise ValueError("state must be (B, 29), actions (B, >= A, 8)")
        if (model is None) == (next_obs_tape is None):
            raise ValueError("exactly one of model / next_obs_tape")
        if model is True:
            model = _lib.AntModel.default()
        if next_obs_tape is not None:
            _chk(next_obs_tape, torch.float64, "next_obs_tape", dev)
            if tuple(next_obs_tape.shape) != (B, A, 29):
                raise ValueError(f"next_obs_tape must be ({B}, {A}, 29)")
        if layout is None:
            layout = "soa" if B >= 4096 else "rows"
        if status is None:
            status = torch.zeros(B, dtype=torch.int32, device=dev)
        states = aout = None
        if want_rows:
            if layout == "soa":
                states = torch.zeros(A + 1, 29, B, dtype=torch.float64, device=dev).permute(2, 0, 1)
                aout = torch.zeros(A, 8, B, dtype=torch.float64, device=dev).permute(2, 0, 1)
            else:
                states = torch.zeros(B, A + 1, 29, dtype=torch.float64, device=dev)
                aout = torch.zeros(B, A, 8, dtype=torch.float64, device=dev)
        steps = torch.zeros(B, dtype=torch.int32, device=dev)
        g, gp = _dbl(np.asarray(desired_goal, dtype=np.float64)[:2])
        sl = _lib.Strides(*states.stride()) if want_rows else None
        al = _lib.Strides(*aout.stride()) if want_rows else None
        check(self._h, lib().ditree_ant_rollout(self._h, C.byref(model) if model is not None else None, _ptr(state), _ptr(actions),
                                                 actions.shape[1] * 8, _ptr(next_obs_tape), A * 29, _ptr(status), B, A, gp,
                                                 float(goal_factor) * float(s_global), float(ball_radius), float(s_global),
                                                 _ptr(states), C.byref(sl) if sl is not None else None, _ptr(aout),
                                                 C.byref(al) if al is not None else None, _ptr(steps), self.stream), "ant_rollout")
        return status, states, aout, steps

    # ------------------------------------------------------------------ obstacle ahead
    def obstacle_ahead(self, state):
        """planners/RRT.py:61-81 for a batch of states (B, >=3) f64 on the uploaded maze -> (B,) u8."""
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        B = state.shape[0]
        out = torch.empty(B, dtype=torch.uint8, device=dev)
        check(self._h, lib().ditree_obstacle_ahead(self._h, _ptr(state), state.shape[1] if B else 6, B, _ptr(out),
                                                    self.stream), "obstacle_ahead")
        return out

    # ------------------------------------------------------------------ online plan following
    def follow_plan(self, state, actions, action_idx, path_xy, known, true_maze, scanned, goal_xy, dt, scan_time):
        """run_scenarios_with_lidar_DiTree.py:470-506 in one launch (see include/ditree.h).  All tensors on the
        device: state (6) f64 [in/out], actions (n, 2) f32, path_xy (P, 2) f32, known / scanned (R, C) f32
        [in/out], true_maze (R, C) f32.  Returns (executed (k, 6) f64, event, next action index, obstacle index)."""
        dev = self.device
        _chk(state, torch.float64, "state", dev)
        _chk(actions, torch.float32, "actions", dev)
        _chk(path_xy, torch.float32, "path_xy", dev)
        for nm, t in (("known", known), ("true_maze", true_maze), ("scanned", scanned)):
            _chk(t, torch.float32, nm, dev)
            if tuple(t.shape) != tuple(known.shape):
                raise ValueError(f"{nm}: maze shapes differ")
        n = actions.shape[0]
        executed = torch.zeros(max(n - int(action_idx), 0), 6, dtype=torch.float64, device=dev)
        result = torch.zeros(4, dtype=torch.int32, device=dev)
        g, gp = _dbl(np.asarray(goal_xy, dtype=np.float64)[:2])
        check(self._h, lib().ditree_follow_plan(self._h, _ptr(state), _ptr(actions), n, int(action_idx), _ptr(path_xy),
                                                 path_xy.shape[0], _ptr(known), _ptr(true_maze), _ptr(scanned), gp,
                                                 float(dt), float(scan_time), _ptr(executed), _ptr(result),
                                                 self.stream), "follow_plan")
        event, nxt, obstacle, _ = [int(v) for v in result.cpu().numpy()]
        return executed[: nxt - int(action_idx)], event, nxt, obstacle

    # ------------------------------------------------------------------ lidar
    def lidar_scan(self, poses, maze_dev, want_visited=True):
        dev = self.device
        _chk(poses, torch.float64, "poses", dev)
        _chk(maze_dev, torch.float32, "maze", dev)
        B = poses.shape[0]
        rows, cols = maze_dev.shape
        dist = torch.empty(B, LIDAR_RAYS, dtype=torch.float64, device=dev)
        ends = torch.empty(B, LIDAR_RAYS, 2, dtype=torch.float64, device=dev)
        hit = torch.empty(B, LIDAR_RAYS, dtype=torch.uint8, device=dev)
        vis = torch.empty(B, rows, cols, dtype=torch.uint8, device=dev) if want_visited else None
        check(self._h, lib().ditree_lidar_scan(self._h, _ptr(poses), B, _ptr(maze_dev), rows, cols, _ptr(dist),
                                                _ptr(ends), _ptr(hit), _ptr(vis), self.stream), "lidar_scan")
        return dist, ends, hit, vis


# ---------------------------------------------------------------------- denoiser front end
def _ctx_load_weights(self, blob, manifest):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    check(self._h, lib().ditree_load_weights(self._h, C.c_void_p(blob.ctypes.data), blob.size,
                                             manifest.encode(), self.stream), "load_weights")
    self.weights_owner = None               # NoisePredNet.bind records itself


def _ctx_denoise_reserve(self, max_batch, precision=_lib.PREC_BF16):
    check(self._h, lib().ditree_denoise_reserve(self._h, int(max_batch), int(precision)), "denoise_reserve")


def _ctx_denoise_status(self, clear=True):
    """Layers of the f16 instantiations (PREC_F16X3 / PREC_F16) that were handed activations beyond +-65504 since the last
    clear, as state-dict names; [] for a healthy network and always for bf16 / f32.  Waits for the stream."""
    n = C.c_int32(0)
    buf = C.create_string_buffer(16384)
    check(self._h, lib().ditree_denoise_status(self._h, C.byref(n), buf, len(buf), int(bool(clear)), self.stream), "denoise_status")
    return buf.value.decode().split("\n") if n.value else []


def _ctx_check_range(self):
    """Raise when the f16 range guard fired: the clamped activations make the returned actions wrong, not approximate."""
    layers = self.denoise_status(clear=True)
    if layers:
        self.raise_range_error(layers)


def _ctx_raise_range_error(self, layers):
    if True:
        raise _lib.DitreeError(
            "f16 range guard: activations beyond +-65504 were clamped in " + ", ".join(layers[:6]) +
            (f" and {len(layers) - 6} more layers" if len(layers) > 6 else "") +
            ".  This checkpoint does not fit the f16x3 / f16 instantiation: bind it with precision=PREC_BF16X3 "
            "(f32 exponent range, 16 significand bits) or PREC_F32.")


def _ctx_denoise(self, noise, local_map, cond, t0=None, dt=None, act_norm=None, want_actions=True, check_range=True):
    """noise (B,P,D) f32, local_map (B,n,n) f32 scaled to {-1,1}, cond (B,G) f32 [device].
    Returns actions (B,P,D) f64 when want_actions else the normalised sample x_K (f32).  check_range: wait for the call
    and raise DitreeError when a layer left the f16 range (no-op cost for bf16 / f32 instantiations apart from the wait;
    pass False inside a throughput loop and call ``check_range()`` once at its end)."""
    dev = self.device
    _chk(noise, torch.float32, "noise", dev)
    _chk(local_map, torch.float32, "local_map", dev)
    _chk(cond, torch.float32, "cond", dev)
    B = noise.shape[0]
    self._check_denoiser_shapes(noise, local_map, cond)
    t0 = np.zeros(1, dtype=np.float32) if t0 is None else t0
    dt = np.ones(1, dtype=np.float32) if dt is None else dt
    t0a, t0p = _flt(t0)
    dta, dtp = _flt(dt)
    an, anp = _dbl(CAR_NORM[12:16] if act_norm is None else act_norm)
    if an.size != 2 * noise.shape[2]:
        raise ValueError(f"act_norm: need mu[{noise.shape[2]}], sigma[{noise.shape[2]}]")
    actions = torch.empty(noise.shape, dtype=torch.float64, device=dev) if want_actions else None
    xout = None if want_actions else torch.empty_like(noise)
    check(self._h, lib().ditree_denoise(self._h, _ptr(noise), _ptr(local_map), _ptr(cond), B, len(t0a), t0p, dtp,
                                        anp, _ptr(actions), _ptr(xout), self.stream), "denoise")
    if check_range:
        self.check_range()
    return actions if want_actions else xout


def _ctx_denoise_ddpm(self, noise, step_noise, local_map, cond, timesteps, coef, act_norm=None, want_actions=True, check_range=True):
    """The sampler's DDPM branch on the device (include/ditree.h ditree_denoise_ddpm): noise (B, P, D) f32 = x_K, step_noise
    (B, K, P, D) f32 = the z of every reverse step, timesteps (K,) / coef (K, 5) from ``ddpm.ddpm_tables``."""
    dev = self.device
    for nm_, t in (("noise", noise), ("step_noise", step_noise), ("local_map", local_map), ("cond", cond)):
        _chk(t, torch.float32, nm_, dev)
    self._check_denoiser_shapes(noise, local_map, cond)
    B, P, D = noise.shape
    ts, tsp = _flt(timesteps)
    cf, cfp = _flt(coef)
    K = len(ts)
    if cf.shape != (K, 5) or tuple(step_noise.shape) != (B, K, P, D):
        raise ValueError(f"coef must be ({K}, 5), step_noise ({B}, {K}, {P}, {D})")
    an, anp = _dbl(CAR_NORM[12:16] if act_norm is None else act_norm)
    if an.size != 2 * D:
        raise ValueError(f"act_norm: need mu[{D}], sigma[{D}]")
    actions = torch.empty(noise.shape, dtype=torch.float64, device=dev) if want_actions else None
    xout = None if want_actions else torch.empty_like(noise)
    check(self._h, lib().ditree_denoise_ddpm(self._h, _ptr(noise), _ptr(step_noise), _ptr(local_map), _ptr(cond), B, K, tsp, cfp, anp,
                                             _ptr(actions), _ptr(xout), self.stream), "denoise_ddpm")
    if check_range:
        self.check_range()
    return actions if want_actions else xout


def _ctx_denoise_eval(self, sample, local_map, cond, timestep, reuse_encoder=False, check_range=True):
    """One raw network evaluation net(sample, map, timestep, cond) -> (B,P,D) f32 (the DDPM branch's model call)."""
    dev = self.device
    _chk(sample, torch.float32, "sample", dev)
    _chk(local_map, torch.float32, "local_map", dev)
    _chk(cond, torch.float32, "cond", dev)
    self._check_denoiser_shapes(sample, local_map, cond)
    out = torch.empty_like(sample)
    check(self._h, lib().ditree_denoise_eval(self._h, _ptr(sample), _ptr(local_map), _ptr(cond), sample.shape[0],
                                             float(timestep), int(bool(reuse_encoder)), _ptr(out), self.stream), "denoise_eval")
    if check_range:
        self.check_range()
    return out


def _ctx_round_stats(self):
    """{calls, waves, quantum} of the last early-exit round on this ctx (include/ditree.h ditree_round_stats)."""
    d = (C.c_int32 * 4)()
    check(self._h, lib().ditree_round_stats(self._h, d), "round_stats")
    return {"denoiser_calls": int(d[0]), "tile_waves": int(d[1]), "candidates_per_wave": int(d[2])}


def _ctx_denoise_dims(self):
    """(pred_horizon, action_dim, local_map_size, obs-cond width, map embedding) of the loaded denoiser."""
    d = (C.c_int32 * 5)()
    check(self._h, lib().ditree_denoise_dims(self._h, d), "denoise_dims")
    return tuple(int(v) for v in d)


def _ctx_check_denoiser_shapes(self, sample, local_map, cond):
    """The library strides its inputs by the loaded network's dimensions: reject tensors of any other shape."""
    P, D, lm, G, _ = self.denoise_dims()
    B = sample.shape[0]
    if tuple(sample.shape) != (B, P, D):
        raise ValueError(f"sample / noise must be ({B}, {P}, {D}) for the loaded denoiser, got {tuple(sample.shape)}")
    if tuple(local_map.shape) != (B, lm, lm):
        raise ValueError(f"local_map must be ({B}, {lm}, {lm}), got {tuple(local_map.shape)}")
    if tuple(cond.shape) != (B, G):
        raise ValueError(f"cond must be ({B}, {G}), got {tuple(cond.shape)}")


def _ctx_debug_read(self, name, B, capacity=1 << 26):
    out = torch.empty(capacity, dtype=torch.float32, device=self.device)
    dims = (C.c_int32 * 3)()
    check(self._h, lib().ditree_denoise_debug_read(self._h, name.encode(), B, _ptr(out), capacity, dims,
                                                   self.stream), "debug_read")
    n = dims[0] * dims[1] * dims[2]
    return out[:n].reshape(dims[0], dims[1], dims[2]).clone()


Context.load_weights = _ctx_load_weights
Context.denoise_reserve = _ctx_denoise_reserve
Context.denoise = _ctx_denoise
Context.denoise_eval = _ctx_denoise_eval
Context.denoise_ddpm = _ctx_denoise_ddpm
Context.debug_read = _ctx_debug_read
Context.denoise_dims = _ctx_denoise_dims
Context.round_stats = _ctx_round_stats
Context.denoise_status = _ctx_denoise_status
Context.check_range = _ctx_check_range
Context.raise_range_error = _ctx_raise_range_error
Context._check_denoiser_shapes = _ctx_check_denoiser_shapes


def _ctx_profile(self, enable=True):
    """False / 0: off; True / 1: bracket every MFMA launch with events; 2: only runs of the dominant (halo) kernel,
    one event pair per run of back-to-back launches (what a timed region can carry without slowing down)."""
    check(self._h, lib().ditree_profile(self._h, int(enable)), "profile")


def profile_kind_names(precision=None):
    """Names of the three MFMA kernel kinds as rocprofv3 prints them (profiles/*_kernel_stats.csv) for a denoiser
    instantiation ("bf16", "f32", "f16x3", "bf16x3", "f16", or a PREC_* value): kind 0 the k = 3 conv kernel, 1 the other
    dense layers, 2 the encoder's implicit Conv2d."""
    name = {v: k for k, v in _lib.PREC_NAMES.items()}.get(precision, precision)
    et = 1 if name in ("f16", "f16x3") else 0
    if name == "f32":
        return ("(none: f32 runs every layer on conv_gemm_kernel)", "conv_gemm_kernel<1, false>", "conv_gemm_kernel<1, true>")
    if name in ("f16x3", "bf16x3"):
        return (f"conv3_halo16x3_kernel<{et}>", f"gemm16_kernel<{et}, true>", f"conv2d_small_kernel<{et}, true>")
    if name in ("f16", "bf16"):
        return (f"conv3_halo16_kernel<{et}, false>", f"gemm16_kernel<{et}, false> + conv_gemm_kernel<{2 if et else 0}, false>",
                f"conv2d_small_kernel<{et}, false>")
    return ("conv3_halo16*_kernel", "gemm16_kernel + conv_gemm_kernel", "conv2d_small_kernel")


def _ctx_profile_read(self, precision=None):
    """-> {rocprof kernel name: dict(ms, launches, flops)} for the MFMA kernels since profile(True)."""
    ms, n, fl = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
    check(self._h, lib().ditree_profile_read(self._h, ms, n, fl), "profile_read")
    names = profile_kind_names(precision)
    return {names[k]: dict(ms=ms[k], launches=n[k], flops=fl[k]) for k in range(3)}


Context.profile = _ctx_profile
Context.profile_read = _ctx_profile_read


_DEFAULT = {}


def default_context(device: int | None = None) -> Context:
    """Process-wide Context per device (the facades share it, like the reference shares one env)."""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _DEFAULT:
        _DEFAULT[device] = Context(device)
    return _DEFAULT[device]
