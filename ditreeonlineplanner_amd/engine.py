"""Device-resident tree + round driver of the DiTree expansion engine.

The reference keeps a Python list of ``Node`` objects (planners/base_planner.py:24-34)
and expands one candidate at a time (planners/RRT.py:131-219).  Here the tree is a
struct of arrays in HBM and a *round* expands B candidates against the tree snapshot:
nearest node -> n_chunks x [local map -> conditioning -> denoiser -> 8-step rollout with
goal / collision tests] -> accept in candidate order.  B = 1 is the reference loop.

With ``world_size > 1`` (one process per GPU, torch.distributed / RCCL) each rank expands
a contiguous block of the round's candidates and the candidate records are all-gathered
before the (replicated, deterministic) accept step, so every rank holds the same tree.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Round, RoundParams, Tree, check, lib
from .ops import CAR_NORM, Context, _dbl, _flt, _ptr, local_axis

CNT_NODES, CNT_GOAL, CNT_LATCH, CNT_ITERS, CNT_CANDS, CNT_STICKY, CNT_OVERFLOW, CNT_PHANTOM = range(8)


class DeviceTree:
    """planners/base_planner.py:24-34 ``Node`` as a struct of arrays in HBM (include/ditree.h ditree_tree).  ``state_dim`` /
    ``action_dim``: 6 / 2 for the car, 29 / 8 for the ant; ``hist``: keep the last three rows of every node's edge (what the
    first sampler call of a child sees with obs_history 3 -- the ant)."""

    def __init__(self, ctx: Context, capacity: int, n_chunks: int, A: int, track_obstacle_ahead: bool = False,
                 state_dim: int = 6, action_dim: int = 2, hist: bool = False):
        dev = ctx.device
        self.capacity, self.n_chunks, self.A = capacity, n_chunks, A
        self.S, self.D = S, D = int(state_dim), int(action_dim)
        f64, i32, u8 = torch.float64, torch.int32, torch.uint8
        self.state = torch.zeros(capacity, S, dtype=f64, device=dev)
        self.xy = torch.zeros(capacity, 2, dtype=f64, device=dev)
        self.parent = torch.full((capacity,), -1, dtype=i32, device=dev)
        self.last_action = torch.zeros(capacity, D, dtype=f64, device=dev)
        self.has_prev = torch.zeros(capacity, dtype=u8, device=dev)
        self.num_visit = torch.zeros(capacity, dtype=i32, device=dev)
        self.edge_states = torch.zeros(capacity, n_chunks * (A + 1), S, dtype=f64, device=dev)
        self.edge_actions = torch.zeros(capacity, n_chunks * A, D, dtype=f64, device=dev)
        self.edge_nstates = torch.zeros(capacity, dtype=i32, device=dev)
        self.edge_nactions = torch.zeros(capacity, dtype=i32, device=dev)
        self.counters = torch.zeros(8, dtype=i32, device=dev)
        # RRT.py:202-205: per-node check_obstacle_ahead flag, kept only for run_type > 0
        self.obstacle_ahead = torch.zeros(capacity, dtype=u8, device=dev) if track_obstacle_ahead else None
        # rank holding each node's edge rows (sharded rounds keep trajectories on the producing rank); -1 = every rank
        self.edge_owner = torch.full((capacity,), -1, dtype=i32, device=dev)
        self.hist = torch.zeros(capacity, 3, S, dtype=f64, device=dev) if hist else None
        self.hist_n = torch.zeros(capacity, dtype=i32, device=dev) if hist else None
        self.desc = Tree(capacity, n_chunks, A, S, D, *[None if t is None else t.data_ptr() for t in (
            self.state, self.xy, self.parent, self.last_action, self.has_prev, self.num_visit,
            self.edge_states, self.edge_actions, self.edge_nstates, self.edge_nactions, self.obstacle_ahead,
            self.edge_owner, self.hist, self.hist_n, self.counters)])
        self.record_doubles = int(lib().ditree_record_doubles(C.byref(self.desc)))
        self.n_nodes_host = 0

    def reset(self, start_state):
        s = torch.as_tensor(np.asarray(start_state, dtype=np.float64), device=self.state.device)
        self.state[0] = s
        self.xy[0] = s[:2]
        self.parent[0] = -1
        self.last_action[0] = 0
        self.has_prev[0] = 0
        self.num_visit.zero_()
        self.edge_nstates[0] = 0
        self.edge_nactions[0] = 0
        if self.obstacle_ahead is not None:
            self.obstacle_ahead.zero_()
        if self.hist is not None:                       # RRT.py:146: the root's child sees curr_state[None, None, :]
            self.hist[0].zero_()
            self.hist[0, 2] = s
            self.hist_n[0] = 1
        self.counters.zero_()
        self.counters[CNT_NODES] = 1
        self.counters[CNT_GOAL] = -1
        self.counters[CNT_PHANTOM] = -1
        self.n_nodes_host = 1

    def read_counters(self):
        c = self.counters.cpu().numpy()
        self.n_nodes_host = int(c[CNT_NODES])
        return c


class RoundBuffers:
    def __init__(self, ctx: Context, B: int, n_chunks: int, A: int, state_dim: int = 6, action_dim: int = 2, hist: bool = False,
                 record_doubles: int = _lib.RECORD_DOUBLES):
        dev = ctx.device
        self.B = B
        self.S, self.D, self.with_hist, self.R = int(state_dim), int(action_dim), bool(hist), int(record_doubles)
        S, D = self.S, self.D
        f64, i32 = torch.float64, torch.int32
        self.parent = torch.zeros(B, dtype=i32, device=dev)
        self.status = torch.zeros(B, dtype=i32, device=dev)
        self.chunks_run = torch.zeros(B, dtype=i32, device=dev)
        self.end_state = torch.zeros(B, S, dtype=f64, device=dev)
        self.states = torch.zeros(B, n_chunks, A + 1, S, dtype=f64, device=dev)
        self.actions = torch.zeros(B, n_chunks, A, D, dtype=f64, device=dev)
        self.chunk_steps = torch.zeros(B, n_chunks, dtype=i32, device=dev)
        self.node_id = torch.full((B,), -1, dtype=i32, device=dev)
        # sharded rounds only (allocated on first use): exchanged records (96 bytes for the car) and what they carry beyond
        # the SoA above
        self.records = self.last_action = self.first_action = self.hist = self.hist_n = None

    def ensure_exchange(self, rows):
        if self.records is None or self.records.shape[0] < rows:
            dev = self.parent.device
            n = max(rows, self.B)
            self.records = torch.zeros(rows, self.R, dtype=torch.float64, device=dev)
            self.last_action = torch.zeros(n, self.D, dtype=torch.float64, device=dev)
            self.first_action = torch.zeros(n, self.D, dtype=torch.float64, device=dev)
            if self.with_hist:
                self.hist = torch.zeros(n, 3, self.S, dtype=torch.float64, device=dev)
                self.hist_n = torch.zeros(n, dtype=torch.int32, device=dev)

    def fields(self):
        return (self.parent, self.status, self.chunks_run, self.end_state, self.states, self.actions,
                self.chunk_steps, self.node_id)

    def desc(self, lo=0, n=None, own=None, shard=0):
        """Round descriptor of rows [lo, lo + n).  ``own`` = (own_lo, own_n) in the descriptor's row numbering and
        ``shard`` = candidates per rank for a sharded round (default: every row was produced here)."""
        n = self.B - lo if n is None else n
        ptrs = [t[lo:lo + n].data_ptr() if n > 0 else t.data_ptr() for t in self.fields()]
        la = fa = hs = hn = None
        if shard and self.last_action is not None:
            la, fa = self.last_action[lo:].data_ptr(), self.first_action[lo:].data_ptr()
            if self.hist is not None:
                hs, hn = self.hist[lo:].data_ptr(), self.hist_n[lo:].data_ptr()
        own_lo, own_n = (0, n) if own is None else own
        return Round(n, *ptrs, la, fa, own_lo, own_n, shard, hs, hn)


def allgather_round_fields(fields, per, rank, world, group=None):
    """One fixed-stride all-gather per candidate-record field.

    Every field is a tensor whose leading axis is the round's candidate index; rank r produced
    rows [r*per, (r+1)*per).  After the call all ranks hold all rows, so the replicated,
    deterministic accept step builds the same tree everywhere.  With the nccl backend (RCCL) the
    gather runs device-to-device over xGMI; with gloo (CPU tests, or several ranks sharing one
    GPU in a test) CUDA tensors are staged through the host.
    """
    import torch.distributed as dist
    backend = dist.get_backend(group)
    pairs = []
    for t in fields:
        if t.shape[0] < per * world:
            raise ValueError("round buffers must hold ceil(B / world) * world candidates")
        flat = t[: per * world]
        pairs.append((flat, flat[rank * per:(rank + 1) * per]))
    if backend == "gloo":
        for flat, mine in pairs:
            if flat.is_cuda:
                host = torch.empty(flat.shape, dtype=flat.dtype)
                dist.all_gather_into_tensor(host, mine.cpu().contiguous(), group=group)
                flat.copy_(host)
            else:
                dist.all_gather_into_tensor(flat, mine.clone(), group=group)
        return
    # RCCL: one collective per field -- ONE per round in the default exchange (the packed records), seven with
    # DITREE_GATHER_EDGES=1 (measured 0.4 ms of a 26 ms round on one MI355X).  With
    # DITREE_COALESCE=1 they are issued as one grouped launch through torch's coalescing manager
    # (no measurable gain on one GPU, so the plain, universally supported calls are the default).
    import os
    sends = [mine.clone() for _, mine in pairs]
    cm = getattr(dist, "_coalescing_manager", None) if os.environ.get("DITREE_COALESCE", "0") == "1" else None
    if cm is not None and not _COALESCE_BROKEN[0]:
        try:
            with cm(group=group, device=pairs[0][0].device, async_ops=False):
                for (flat, _), snd in zip(pairs, sends):
                    dist.all_gather_into_tensor(flat, snd, group=group)
            return
        except Exception:                              # pragma: no cover - depends on the torch build
            _COALESCE_BROKEN[0] = True
    for (flat, _), snd in zip(pairs, sends):
        dist.all_gather_into_tensor(flat, snd, group=group)


_COALESCE_BROKEN = [False]


def default_shard():
    """(rank, world_size, process_group) a facade shards over when the caller names none.  Sharding over the default
    torch.distributed group is OPT-IN -- DITREE_SHARD_DEFAULT_GROUP=1, which `python -m ditreeonlineplanner_amd.run` sets when
    it joins the group for a driver script -- because it is only sound when every rank plans the SAME scenario with identical
    RNG streams: a data-parallel harness whose ranks plan different scenarios would dead-lock or mix trees in the per-round
    collectives.  Everything else gets (0, 1, None); explicit ``rank`` / ``world_size`` / ``process_group`` always win."""
    import os
    import torch.distributed as dist
    if os.environ.get("DITREE_SHARD_DEFAULT_GROUP", "0") == "1" and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), None
    return 0, 1, None


class ExpansionEngine:
    """Batched RRT expansion on one GPU (optionally one shard of a multi-GPU round)."""
    STATE_DIM, ACTION_DIM, HIST = 6, 2, False

    def __init__(self, ctx: Context, maze, start_state, goal_state, edge_length=64, action_horizon=8,
                 pred_horizon=64, local_map_size=20, local_map_scale=0.2, s_global=1.0, batch=1024,
                 capacity=65536, k_steps=1, emulate_sticky_done=True, norm=CAR_NORM, rank=0, world_size=1,
                 process_group=None, early_exit=False, run_type=0, goal_scale=None, prop_duration=None):
        self.ctx = ctx
        self.maze = np.asarray(maze, dtype=np.float32)
        # planners/RRT.py:26,149-152: per-visit edge lengths; the tree's edge capacity is the longest entry
        self.schedule = [int(edge_length)] if prop_duration is None else [int(v) for v in prop_duration]
        if not self.schedule or any(v < action_horizon or v % action_horizon for v in self.schedule) or len(self.schedule) > 16:
            raise ValueError("prop_duration: 1..16 edge lengths, each a positive multiple of action_horizon")
        edge_length = max(self.schedule)
        self.H, self.A, self.P = edge_length, action_horizon, pred_horizon
        self.n_chunks = edge_length // action_horizon
        self.lm_n, self.lm_scale, self.s_global = local_map_size, local_map_scale, s_global
        # divisor of the goal offset inside tanh (fm_policy.py:125-143 uses the SAMPLER's local_map_size)
        self.goal_scale = float(local_map_size if goal_scale is None else goal_scale)
        self.batch = batch
        self.k_steps = k_steps
        self.sticky = int(bool(emulate_sticky_done))
        self.early_exit = int(bool(early_exit))    # skip later chunks of collided / finished candidates
        self.norm = np.ascontiguousarray(CAR_NORM if norm is None else norm, dtype=np.float64)
        self.rank, self.world, self.pg = rank, world_size, process_group
        self.ddpm = None                    # (timesteps (K,), coef (K, 5)): the sampler's DDPM branch instead of the flow steps
        self.force_allgather = False        # run the collective even with one rank (exercises the RCCL path on 1 GPU)
        self.exchange_events = None         # list -> one (start, end) event pair per round around pack + all-gather + unpack
        self.run_type = int(run_type)
        self.init_main_path = None          # (P, >=2) reference path of an earlier plan (run_type > 0)
        self.tree = DeviceTree(ctx, capacity, self.n_chunks, self.A, track_obstacle_ahead=self.run_type > 0,
                               state_dim=self.STATE_DIM, action_dim=self.ACTION_DIM, hist=self.HIST)
        # sharded rounds exchange ceil(batch / world) fixed slots per rank: size the round buffers for whole slots, so a batch
        # that does not divide by the rank count is fine (the last rank's tail slots stay unused)
        per = (batch + world_size - 1) // world_size
        self.rb = RoundBuffers(ctx, per * world_size, self.n_chunks, self.A, state_dim=self.STATE_DIM,
                               action_dim=self.ACTION_DIM, hist=self.HIST, record_doubles=self.tree.record_doubles)
        self._budget = self._budget_parent = None
        if len(self.schedule) > 1:
            self._budget = torch.zeros(batch, dtype=torch.int32, device=ctx.device)
            self._budget_parent = torch.zeros(batch, dtype=torch.int32, device=ctx.device)
            self._sched_chunks = (C.c_int32 * len(self.schedule))(*[v // action_horizon for v in self.schedule])
        self.axis = local_axis(local_map_size, local_map_scale)
        from .common.fm_utils import get_timesteps
        t0, dt = get_timesteps("exp", k_steps, 4.0)
        self.t0, self.dt = t0.numpy().copy(), dt.numpy().copy()
        ctx.upload_maze(self.maze, owner=self)
        self.reset(start_state, goal_state)

    # ------------------------------------------------------------------ state
    def reset(self, start_state, goal_state):
        self.start_state = np.asarray(start_state, dtype=np.float64).copy()
        self.goal_state = np.asarray(goal_state, dtype=np.float64).copy()
        if self.start_state.shape != (self.STATE_DIM,):
            raise ValueError(f"start_state must have {self.STATE_DIM} elements")
        self._derive_env_goal()
        self.tree.reset(self.start_state)
        self.generation = getattr(self, "generation", 0) + 1      # consumers caching per-tree data (node_list) key on it

    def _derive_env_goal(self):
        H, W = self.maze.shape
        # planner.reset -> env.reset(options): env.goal = centre of the goal cell (car_env.py:189-201,225-226)
        gi = np.floor((H / 2 - self.goal_state[1]) / 1.0)
        gj = np.floor((self.goal_state[0] + W / 2) / 1.0)
        self.env_goal = np.array([(gj + 0.5) * 1.0 - W / 2, H / 2 - (gi + 0.5) * 1.0])

    def update_maze(self, maze):
        self.maze = np.asarray(maze, dtype=np.float32)
        self.ctx.upload_maze(self.maze, owner=self)

    def ensure_maze(self):
        """The ctx's device maze must be THIS engine's before any launch that reads it (another planner, or a
        check_collision call, may have uploaded its own since)."""
        if self.ctx.maze_owner is not self:
            self.ctx.upload_maze(self.maze, owner=self)

    def shard(self, B):
        """Contiguous candidate block of this rank."""
        per = (B + self.world - 1) // self.world
        lo = min(self.rank * per, B)
        hi = min(lo + per, B)
        return lo, hi, per

    def agree(self, flag: bool) -> bool:
        """Rank 0's decision for every rank (the wall-clock budget of a plan is read on rank 0 only: ranks that disagreed
        about running another round would dead-lock in its collective).  One 4-byte broadcast; identity for one rank."""
        if self.world <= 1:
            return bool(flag)
        import torch.distributed as dist
        gloo = dist.get_backend(self.pg) == "gloo"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if gloo else self.ctx.device)
        src = 0 if self.pg is None else dist.get_global_rank(self.pg, 0)
        dist.broadcast(t, src=src, group=self.pg)
        return bool(int(t.item()))

    def sum_over_ranks(self, value: int, op="sum") -> int:
        if self.world <= 1:
            return int(value)
        import torch.distributed as dist
        gloo = dist.get_backend(self.pg) == "gloo"
        t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if gloo else self.ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.pg)
        return int(t.item())

    # ------------------------------------------------------------------ one round
    def _sampler_schedule(self, rp, keep, step_noise, lo, hi):
        """t0 / dt of the flow steps, or -- self.ddpm set -- the DDPM timesteps, coefficient table and this round's step noise."""
        if self.ddpm is None:
            for name, arr in (("t0", self.t0), ("dt", self.dt)):
                a, p = _flt(arr)
                keep.append(a)
                setattr(rp, name, p)
            rp.K = self.k_steps
            rp.ddpm_coef, rp.step_noise = None, None
            return
        ts, coef = self.ddpm
        a, rp.t0 = _flt(ts)
        b, rp.ddpm_coef = _flt(coef)
        keep += [a, b, step_noise]
        rp.dt = None
        rp.K = len(a)
        if step_noise is None:
            raise ValueError("the DDPM branch needs step_noise (B, n_chunks, K, P, D) f32")
        want = (self.n_chunks, rp.K, self.P, self.ACTION_DIM)
        if tuple(step_noise.shape[1:]) != want or step_noise.dtype != torch.float32 or not step_noise.is_contiguous():
            raise ValueError(f"step_noise must be a contiguous float32 (B, {want[0]}, {want[1]}, {want[2]}, {want[3]}) tensor")
        rp.step_noise = step_noise[lo:hi].data_ptr() if hi > lo else None

    def expand_round(self, samples, cond_goal, noise=None, inject_actions=None, accept=True, step_noise=None):
        """samples (B,6) f64, cond_goal (B,2) f64 [device tensors, all candidates of the round];
        noise (B, n_chunks, P, 2) f32 or inject_actions (B, n_chunks, P, 2) f64 for this round; step_noise
        (B, n_chunks, K, P, 2) f32 with ``self.ddpm`` set (the sampler's DDPM branch)."""
        B = samples.shape[0]
        if B > self.batch:
            raise ValueError(f"round of {B} candidates exceeds engine batch {self.batch}")
        lo, hi, per = self.shard(B)
        n = hi - lo
        self.ensure_maze()
        rp = RoundParams()
        rp.n_nodes = self.tree.n_nodes_host
        rp.samples = samples[lo:hi].data_ptr() if n else None
        rp.cond_goal = cond_goal[lo:hi].data_ptr() if n else None
        rp.noise = noise[lo:hi].data_ptr() if (noise is not None and n) else None
        rp.inject_actions = inject_actions[lo:hi].data_ptr() if (inject_actions is not None and n) else None
        rp.P = self.P
        keep = []
        if self.ddpm is None or noise is None:
            self._flow_only(rp, keep)
        else:
            self._sampler_schedule(rp, keep, step_noise, lo, hi)
        for name, arr, conv in (("norm", self.norm, _dbl), ("goal_xy", self.env_goal, _dbl), ("axis", self.axis, _dbl)):
            a, p = conv(arr)
            keep.append(a)
            setattr(rp, name, p)
        rp.lm_n, rp.lm_size, rp.s_global = self.lm_n, self.goal_scale, float(self.s_global)
        rp.early_exit = self.early_exit
        rp.chunk_budget = None
        if self._budget is not None:
            # every rank derives the budgets of the WHOLE round (a candidate's place in its parent's visit order is global)
            check(self.ctx._h, lib().ditree_chunk_budget(self.ctx._h, C.byref(self.tree.desc), samples.data_ptr(), B,
                                                          self.tree.n_nodes_host, self._sched_chunks, len(self.schedule),
                                                          self._budget_parent.data_ptr(), self._budget.data_ptr(),
                                                          self.ctx.stream), "chunk_budget")
            rp.chunk_budget = self._budget[lo:hi].data_ptr() if n else None
        if n > 0:
            rd = self.rb.desc(lo, n)
            check(self.ctx._h, lib().ditree_expand_round(self.ctx._h, C.byref(self.tree.desc), C.byref(rd),
                                                          C.byref(rp), self.ctx.stream), "expand_round")
        return self._finish_round(B, per, noise is not None, accept)

    def _flow_only(self, rp, keep):
        """An action-tape round of an engine whose sampler is the DDPM branch: no sampler call, the schedule is irrelevant."""
        saved, self.ddpm = self.ddpm, None
        try:
            self._sampler_schedule(rp, keep, None, 0, 0)
        finally:
            self.ddpm = saved

    def _finish_round(self, B, per, used_denoiser, accept):
        """The part of a round behind the expansion: the record exchange (sharded rounds) and the replicated accept."""
        if self.world > 1 or self.force_allgather:
            ev = None
            if self.exchange_events is not None:
                # torch's collective runs on the backend's own stream, but the current stream waits for it (async_op=False):
                # events on the current stream bracket pack + collective + unpack
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            self._allgather_round(B, per)
            if ev is not None:
                ev[1].record()
                self.exchange_events.append(ev)
        self._used_denoiser = bool(used_denoiser)
        if accept:
            return self.accept(B)
        return None

    def _allgather_round(self, B, per):
        """One collective per round: 96-byte candidate records (end state, last / first action, parent, status, chunks run;
        SURVEY.md 8(e)).  The edge trajectories stay on the rank that produced them (tree.edge_owner) and are only moved
        by ``path_to``.  DITREE_GATHER_EDGES=1 restores the round-1 behaviour (every field incl. trajectories)."""
        import os
        if os.environ.get("DITREE_GATHER_EDGES", "0") == "1":
            self._sharded = False
            allgather_round_fields(self.rb.fields()[:-1], per, self.rank, self.world, self.pg)
            return
        self._sharded = True
        rows = per * self.world
        rb = self.rb
        rb.ensure_exchange(rows)
        lo, hi, _ = self.shard(B)
        n = hi - lo
        mine = rb.records[self.rank * per:(self.rank + 1) * per]
        if n > 0:
            rd = rb.desc(lo, n)
            check(self.ctx._h, lib().ditree_round_pack(self.ctx._h, C.byref(self.tree.desc), C.byref(rd), mine.data_ptr(),
                                                        self.ctx.stream), "round_pack")
        allgather_round_fields([rb.records], per, self.rank, self.world, self.pg)
        rd = rb.desc(0, B, shard=per)
        check(self.ctx._h, lib().ditree_round_unpack(self.ctx._h, C.byref(self.tree.desc), C.byref(rd), rb.records.data_ptr(),
                                                      self.ctx.stream), "round_unpack")

    def accept(self, B):
        self.ensure_maze()                        # run_type > 0: the commit kernel evaluates check_obstacle_ahead
        if getattr(self, "_sharded", False) and (self.world > 1 or self.force_allgather):
            lo, hi, per = self.shard(B)
            rd = self.rb.desc(0, B, own=(lo, hi - lo), shard=per)
        else:
            rd = self.rb.desc(0, B)
        check(self.ctx._h, lib().ditree_accept(self.ctx._h, C.byref(self.tree.desc), C.byref(rd), self.sticky,
                                                self.ctx.stream), "accept")
        cnt = self.tree.read_counters()           # one small D2H per round: n_nodes / goal
        if getattr(self, "_used_denoiser", False):
            # f16 range guard: the stream is drained already, one more tiny D2H.  Sharded: a rank whose shard saturated must
            # not raise alone (the others would wait in the next collective until the launcher kills them) -- the ranks agree
            # on the worst count first and then all raise.
            layers = self.ctx.denoise_status(clear=True)
            worst = self.sum_over_ranks(len(layers), op="max")
            if worst:
                if not layers:
                    raise _lib.DitreeError(f"f16 range guard: another rank clamped activations in {worst} layer(s) of this round; "
                                           "bind the checkpoint with precision=PREC_BF16X3 or PREC_F32")
                self.ctx.raise_range_error(layers)
        return cnt

    # ------------------------------------------------------------------ results
    @property
    def goal_node(self):
        g = int(self.tree.counters[CNT_GOAL].item())
        return None if g < 0 else g

    def fallback_node(self):
        """planners/RRT.py:227-254.  run_type 0: among nodes 1.., the one nearest to goal_state xy (arg-min
        reduction on the device: the nearest-node kernel with the goal as the single query).  run_type > 0:
        None when every node has an obstacle ahead; else nearest to the goal with +1e4 on flagged nodes, or --
        with a reference path -- the unflagged node whose nearest path point lies furthest along it."""
        n = self.tree.n_nodes_host
        if n < 2:
            return None
        if self.run_type > 0:
            out = torch.empty(1, dtype=torch.int32, device=self.tree.xy.device)
            g, gp = _dbl(self.goal_state[:2])
            if self.init_main_path is not None:
                pa, pp = _dbl(np.asarray(self.init_main_path)[:, :2])
                P = pa.shape[0]
            else:
                pa, pp, P = None, None, 0
            check(self.ctx._h, lib().ditree_fallback_select(self.ctx._h, C.byref(self.tree.desc), n, gp, pp, P,
                                                             out.data_ptr(), self.ctx.stream), "fallback_select")
            node = int(out.item())
            return None if node < 0 else node
        q = torch.as_tensor(self.goal_state[:2].reshape(1, 2).copy(), device=self.tree.xy.device)
        idx = self.ctx.nn_argmin(q, self.tree.xy[1:n].contiguous(), n_nodes=n - 1)
        return 1 + int(idx[0].item())

    def path_to(self, node):
        """planners/base_planner.py:342-363: float32 path (edge states + node states) and actions."""
        t = self.tree
        parents = t.parent[: t.n_nodes_host].cpu().numpy()
        chain = []
        n = int(node)
        while n != -1:
            chain.append(n)
            n = int(parents[n])
        chain = chain[::-1]
        idx = torch.as_tensor(chain, device=t.state.device, dtype=torch.long)
        st = t.state[idx].cpu().numpy()
        es_t, ea_t = t.edge_states[idx], t.edge_actions[idx]
        ns_t, na_t = t.edge_nstates[idx], t.edge_nactions[idx]
        if self.world > 1 and getattr(self, "_sharded", False):
            # collective (every rank walks the same chain): each edge comes from the rank that expanded it; rows a rank
            # does not hold are zeroed, the sum over ranks is then exact (x + 0 + ... + 0)
            import torch.distributed as dist
            owner = t.edge_owner[idx]
            mine = (owner == self.rank) | ((owner < 0) & (self.rank == 0))
            mask = mine.to(es_t.dtype)
            es_t = es_t * mask[:, None, None]
            ea_t = ea_t * mask[:, None, None]
            ns_t = torch.where(mine, ns_t, torch.zeros_like(ns_t))
            na_t = torch.where(mine, na_t, torch.zeros_like(na_t))
            gloo = dist.get_backend(self.pg) == "gloo"
            bufs = []
            for x in (es_t, ea_t, ns_t, na_t):
                y = x.cpu() if gloo else x.contiguous()
                dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.pg)
                bufs.append(y)
            es_t, ea_t, ns_t, na_t = bufs
        es, ea = es_t.cpu().numpy(), ea_t.cpu().numpy()
        ns, na = ns_t.cpu().numpy(), na_t.cpu().numpy()
        path, actions = [], []
        for k, nd in enumerate(chain):
            if nd != 0:
                path.extend(es[k, : ns[k]])
                actions.extend(ea[k, : na[k]])
            path.append(st[k])
        path = np.array(path, dtype=np.float32) if path else None
        actions = np.array(actions, dtype=np.float32) if actions else None
        return path, actions

    def tree_snapshot(self):
        n = self.tree.n_nodes_host
        return dict(parents=self.tree.parent[:n].cpu().numpy(), states=self.tree.state[:n].cpu().numpy(),
                    counters=self.tree.counters.cpu().numpy())


class AntExpansionEngine(ExpansionEngine):
    """The expansion rounds of BASELINE config 3 (cfgs/antmaze.yaml: 29-d states, 8-d actions, action_horizon 2, edge length
    48, obs_history 3, local map 16 @ 0.8, s_global 4): nearest node -> 24 chunks x [local map -> ant conditioning vector ->
    denoiser -> 2 env steps with the reference's goal (planners/base_planner.py:296-297) and collision tests
    (common/map_utils.py:126-219)] -> accept (planners/RRT.py:195-217), on a tree that also keeps every node's last three
    edge rows (the history a child's first sampler call sees).

    THE ENV STEP is MuJoCo in the reference (third party, no oracle here) and is NOT implemented.  ``dynamics``:
      "tape"   ``expand_round(..., next_obs_tape=(B, n_chunks, A, 29))``: observations given from outside;
      "model"  the build's stand-in crawler model (include/ditree.h ditree_ant_model; not MuJoCo, parity unpinned);
      "host"   ``expand_round(..., step_fn=...)``: the caller's simulator, ``step_fn(chunk, start_states (n, 29), actions
               (n, A, 8), rows (n,)) -> (n, A, 29)`` observations, called once per chunk for the candidates still alive.
    """
    STATE_DIM, ACTION_DIM, HIST = 29, 8, True

    def __init__(self, ctx: Context, maze, start_state, goal_state, desired_goal=None, norm=None, edge_length=48,
                 action_horizon=2, pred_horizon=16, local_map_size=16, local_map_scale=0.8, s_global=4.0, batch=4096,
                 capacity=65536, k_steps=1, rank=0, world_size=1, process_group=None, early_exit=False, goal_scale=None,
                 dynamics="tape", model=None, ball_radius=1.2, goal_factor=0.45):
        if norm is None:
            raise ValueError("AntExpansionEngine: norm = obs_mean[27] + obs_std[27] + act_mean[8] + act_std[8] (metadata/antmaze.pt)")
        if dynamics not in ("tape", "model", "host"):
            raise ValueError("dynamics must be 'tape', 'model' or 'host'")
        self.dynamics = dynamics
        self.model = _lib.AntModel.default() if model is None else model
        self.ball_radius = float(ball_radius)
        self.goal_radius = float(goal_factor) * float(s_global)          # planners/base_planner.py:297
        self._desired_arg = None if desired_goal is None else np.asarray(desired_goal, dtype=np.float64)[:2].copy()
        super().__init__(ctx, maze, start_state, goal_state, edge_length=edge_length, action_horizon=action_horizon,
                         pred_horizon=pred_horizon, local_map_size=local_map_size, local_map_scale=local_map_scale,
                         s_global=s_global, batch=batch, capacity=capacity, k_steps=k_steps, emulate_sticky_done=False,
                         norm=np.asarray(norm, dtype=np.float64), rank=rank, world_size=world_size, process_group=process_group,
                         early_exit=early_exit, run_type=0, goal_scale=goal_scale)
        if self.norm.size != 70:
            raise ValueError("norm: 27 + 27 + 8 + 8 doubles")

    def _derive_env_goal(self):
        # obs['desired_goal'] of the env (the goal cell's centre plus the env's position noise): given by the caller, else
        # the centre of the goal cell in the scaled map
        if self._desired_arg is not None:
            self.env_goal = self._desired_arg.copy()
            return
        H, W = self.maze.shape
        sg = float(self.s_global)
        gi = np.floor((H / 2 * sg - self.goal_state[1]) / sg)
        gj = np.floor((self.goal_state[0] + W / 2 * sg) / sg)
        self.env_goal = np.array([(gj + 0.5) * sg - W / 2 * sg, H / 2 * sg - (gi + 0.5) * sg])

    def _params(self, samples, cond_goal, noise, inject_actions, lo, hi, next_obs_tape=None, cond_out=None, step_noise=None):
        n = hi - lo
        rp = _lib.AntRoundParams()
        rp.n_nodes = self.tree.n_nodes_host
        rp.samples = samples[lo:hi].data_ptr() if n else None
        rp.cond_goal = cond_goal[lo:hi].data_ptr() if n else None
        rp.noise = noise[lo:hi].data_ptr() if (noise is not None and n) else None
        rp.inject_actions = inject_actions[lo:hi].data_ptr() if (inject_actions is not None and n) else None
        rp.P = self.P
        keep = [samples, cond_goal, noise, inject_actions, next_obs_tape, cond_out]
        if self.ddpm is None or noise is None:
            self._flow_only(rp, keep)
        else:
            self._sampler_schedule(rp, keep, step_noise, lo, hi)
        for name, arr, conv in (("norm", self.norm, _dbl),
                                ("desired_goal", self.env_goal, _dbl), ("axis", self.axis, _dbl)):
            a, p = conv(arr)
            keep.append(a)
            setattr(rp, name, p)
        rp.goal_radius, rp.ball_radius = self.goal_radius, self.ball_radius
        rp.lm_n, rp.lm_size, rp.s_global = self.lm_n, self.goal_scale, float(self.s_global)
        rp.dynamics = _lib.ANT_DYN_MODEL if self.dynamics == "model" else _lib.ANT_DYN_TAPE
        rp.next_obs_tape = next_obs_tape[lo:hi].data_ptr() if (next_obs_tape is not None and n) else None
        rp.model = C.pointer(self.model)
        rp.early_exit = self.early_exit
        rp.cond_out = cond_out[lo:hi].data_ptr() if (cond_out is not None and n) else None
        return rp, keep

    def expand_round(self, samples, cond_goal, noise=None, inject_actions=None, accept=True, next_obs_tape=None, step_fn=None,
                     cond_out=None, step_noise=None):
        """samples (B, 29) f64, cond_goal (B, 2) f64, noise (B, n_chunks, P, 8) f32 or inject_actions (B, n_chunks, P, 8) f64
        [device tensors, all candidates of the round]; next_obs_tape (B, n_chunks, A, 29) f64 for dynamics='tape'; step_fn for
        dynamics='host'; cond_out (B, n_chunks, 97) f32: receives every sampler call's conditioning vector (tests)."""
        B = samples.shape[0]
        if B > self.batch:
            raise ValueError(f"round of {B} candidates exceeds engine batch {self.batch}")
        if tuple(samples.shape) != (B, 29) or tuple(cond_goal.shape) != (B, 2):
            raise ValueError("samples must be (B, 29), cond_goal (B, 2)")
        shape = (B, self.n_chunks, self.P, 8)
        for nm, t in (("noise", noise), ("inject_actions", inject_actions)):
            if t is not None and tuple(t.shape) != shape:
                raise ValueError(f"{nm} must be {shape}, got {tuple(t.shape)}")
        if self.dynamics == "tape" and (next_obs_tape is None or tuple(next_obs_tape.shape) != (B, self.n_chunks, self.A, 29)):
            raise ValueError(f"dynamics='tape': next_obs_tape must be ({B}, {self.n_chunks}, {self.A}, 29)")
        if self.dynamics == "host" and step_fn is None:
            raise ValueError("dynamics='host': step_fn is required")
        lo, hi, per = self.shard(B)
        n = hi - lo
        self.ensure_maze()
        rp, keep = self._params(samples, cond_goal, noise, inject_actions, lo, hi, next_obs_tape, cond_out, step_noise)
        if n > 0:
            rd = self.rb.desc(lo, n)
            h, L, t = self.ctx._h, lib(), C.byref(self.tree.desc)
            if self.dynamics != "host":
                check(h, L.ditree_expand_round_ant(h, t, C.byref(rd), C.byref(rp), self.ctx.stream), "expand_round_ant")
            else:
                self._host_stepped_round(rd, rp, lo, n, step_fn)
        del keep
        return self._finish_round(B, per, noise is not None, accept)

    def _host_stepped_round(self, rd, rp, lo, n, step_fn):
        """The caller's simulator between the two halves of every chunk (include/ditree.h ditree_ant_round_begin / _chunk_sample
        / _chunk_step): per chunk one D2H of the alive candidates' start states and actions, one H2D of their observations."""
        h, L, t, st = self.ctx._h, lib(), C.byref(self.tree.desc), self.ctx.stream
        rb = self.rb
        check(h, L.ditree_ant_round_begin(h, t, C.byref(rd), C.byref(rp), st), "ant_round_begin")
        obs = torch.zeros(n, self.A, 29, dtype=torch.float64, device=self.ctx.device)
        for j in range(self.n_chunks):
            check(h, L.ditree_ant_chunk_sample(h, t, C.byref(rd), C.byref(rp), j, st), "ant_chunk_sample")
            alive = torch.nonzero(rb.status[lo:lo + n] == _lib.ST_OK).flatten()
            if alive.numel() == 0:
                break
            rows = alive.cpu().numpy()
            acts = rb.actions[lo:lo + n, j][alive].cpu().numpy()            # (n_alive, A, 8): what chunk_sample just wrote
            start = rb.end_state[lo:lo + n][alive].cpu().numpy()
            o = np.asarray(step_fn(j, start, acts, rows), dtype=np.float64)
            if o.shape != (rows.size, self.A, 29):
                raise ValueError(f"step_fn returned {o.shape}, need {(rows.size, self.A, 29)}")
            obs[alive] = torch.as_tensor(o, device=obs.device)
            check(h, L.ditree_ant_chunk_step(h, t, C.byref(rd), C.byref(rp), j, obs.data_ptr(), st), "ant_chunk_step")
