"""RRT_Planner facade (reference: planners/RRT.py:18-257) -- the expand loop as wide rounds on the GPU.

Constructor kwargs, ``plan() -> (path f32 (P,6)|None, actions f32 (A,2)|None)``, ``reset``,
``update_maze``, ``results`` and ``node_list`` follow the reference.  Extra kwargs:
  ``batch``            candidates per round (default 512 = one full wave of 256-row tiles on the 256 CUs for the `large`
                       denoiser: rounds of 128 and 256 take the same time as each other, multiples of 512 fill whole waves; 1 = the
                       reference's sequential order),
  ``max_candidates``   deterministic budget instead of / in addition to the wall-clock budget,
  ``early_exit``       (default True) later chunks of collided / finished candidates are skipped, as the
                       reference abandons a collided edge; results are identical either way,
  ``prop_duration``    the reference's per-visit edge-length schedule (RRT.py:26,149-152): in a round the candidates of
                       one parent are its visits in candidate order,
  ``rank`` / ``world_size`` / ``process_group``   shard every round over the ranks of a torch.distributed group (default:
                       the initialised default group, i.e. what `torch.distributed.run` set up; one rank otherwise).  All
                       ranks must draw the same samples and noise -- the reference's scripts seed every RNG with 42
                       (run_scenarios.py:86-90) -- and all ranks return the same path.
All of the reference's ``run_type`` values: 0 "Original"; 1 "Original+Ref" (obstacle-ahead flags per node, sampling
biased to the part of ``init_main_path`` behind the obstacle, furthest-along-path fallback); 2 "OM+Ref" (sample
positions drawn from the EDT prior); 3 "OM+LB+Ref" (prior log-blended with the start -> goal Gaussian, refreshed at
every plan); 4 adds the driver's forced re-plan and is the same as 3 inside the planner.
"""
from __future__ import annotations

import random
import time

import numpy as np
import torch

from ..engine import CNT_GOAL, CNT_ITERS, AntExpansionEngine, ExpansionEngine, default_shard
from .base_planner import BasePlanner, Node


class RRT_Planner(BasePlanner):
    def __init__(self, start_state, goal_state, environment, sampler, **kwargs):
        super().__init__(start_state, goal_state, environment, sampler, **kwargs)
        self.kd_tree_dim = 2
        self.goal_sample_rate = 0.15
        self.goal_conditioning_bias = kwargs.get("goal_conditioning_bias", 0.85)
        self.prop_duration_schedule = list(kwargs.get("prop_duration", [64]))
        # plan() runs fused rounds (ditree_expand_round): K flow-matching Euler steps inside the library, or -- a sampler built
        # with policy='diffusion' and a DDPM scheduler of the reference's configuration (run_scenarios.py:157-158) -- the K
        # DDPM reverse steps (fm_policy.py:164-182; ditree_denoise_ddpm).  Any other scheduler object can only drive
        # DiffusionSampler.forward itself: refuse instead of sampling with the wrong rule.
        self._ddpm = None
        if getattr(sampler, "policy", "flow_matching") == "diffusion":
            self._ddpm = sampler.ddpm_tables() if hasattr(sampler, "ddpm_tables") else None
            if self._ddpm is None:
                raise NotImplementedError("RRT_Planner.plan with policy='diffusion' needs a DDPM scheduler with epsilon prediction, "
                                          "clip_sample and fixed_small variance (the reference's configuration); other schedulers "
                                          "are only available through DiffusionSampler.forward")
        elif getattr(sampler, "policy", "flow_matching") != "flow_matching":
            raise NotImplementedError(f"policy={getattr(sampler, 'policy', None)!r}")
        self.offline_time_budget = kwargs.get("offline_time_budget", 60)
        self.plan_count = 0
        self.init_main_path = None
        self.run_type = kwargs.get("run_type", 0)
        self.env.run_type = self.run_type
        self.batch = int(kwargs.get("batch", 512))
        self.max_candidates = kwargs.get("max_candidates", None)
        self.capacity = int(kwargs.get("capacity", 65536))
        lm = self.local_map_size if isinstance(self.local_map_size, (int, float)) else self.local_map_size[0]
        d_rank, d_world, d_pg = default_shard()
        self.world_size = int(kwargs.get("world_size", d_world))
        self.rank = int(kwargs.get("rank", d_rank if self.world_size == d_world else 0))
        self.process_group = kwargs.get("process_group", d_pg)
        if self.is_ant:
            self._init_ant_engine(sampler, kwargs)
            return
        self._engine = ExpansionEngine(
            self.ctx, self.maze, self.start_node.state, self.goal_state, edge_length=max(self.prop_duration_schedule),
            prop_duration=self.prop_duration_schedule,
            action_horizon=self.action_horizon, pred_horizon=getattr(sampler, "pred_horizon", 64),
            local_map_size=int(lm), local_map_scale=self.local_map_scale, s_global=self.s_global, batch=self.batch,
            capacity=self.capacity, k_steps=getattr(sampler, "num_diffusion_iters", 1),
            norm=getattr(sampler, "norm", None) if getattr(sampler, "norm", None) is not None else None,
            emulate_sticky_done=kwargs.get("emulate_sticky_done", True), early_exit=kwargs.get("early_exit", True),
            run_type=self.run_type, goal_scale=getattr(sampler, "local_map_size", None),
            rank=self.rank, world_size=self.world_size, process_group=self.process_group)
        self._engine.env_goal = np.asarray(self.env.goal, dtype=np.float64)
        self._engine.ddpm = self._ddpm
        from concurrent.futures import ThreadPoolExecutor
        self._draw_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ditree-draw")

    # ------------------------------------------------------------------ ant (BASELINE config 3)
    def _init_ant_engine(self, sampler, kwargs):
        """env_id = 'antmaze' (cfgs/antmaze.yaml, run_scenarios.py:123-132,225-233): 29-d states, 8-d actions, action_horizon 2,
        local map 16 @ 0.8, global_map_scale 4.  Everything of the expand loop runs on the device -- sampling glue, denoiser,
        the reference's collision and goal tests on every observation, accept -- EXCEPT the env step, which is MuJoCo in the
        reference and is not part of this build: ``ant_dynamics``
          "host" (default)  the planner steps the CALLER's env (``environment.ant_env.set_state(qpos, qvel)`` +
                            ``environment.step(action)`` per candidate and step, planners/base_planner.py:278-279,290) between
                            the two halves of every chunk,
          "model"           the build's stand-in crawler model on the device (NOT MuJoCo; parity unpinned),
          "tape"            observations from ``next_obs_tape_fn(first_candidate, B) -> (B, n_chunks, A, 29)`` (tests)."""
        if self.run_type != 0:
            raise NotImplementedError("antmaze: run_type 0 (the reference's check_obstacle_ahead / sampling maps are the car's)")
        if len(self.prop_duration_schedule) != 1:
            raise NotImplementedError("antmaze: a single prop_duration entry")
        self.ant_dynamics = kwargs.get("ant_dynamics", "host")
        self._tape_fn = kwargs.get("next_obs_tape_fn")
        lm = self.local_map_size if isinstance(self.local_map_size, (int, float)) else self.local_map_size[0]
        norm = getattr(sampler, "norm", None)
        if norm is None:                                        # a non-network sampler: the packaged copy of metadata/antmaze.pt
            from ..policies.fm_policy import load_metadata
            md = load_metadata("antmaze")
            norm = np.concatenate([md["Observations_mean"], md["Observations_std"], md["Actions_mean"], md["Actions_std"]])
        desired = kwargs.get("desired_goal")
        out = getattr(self, "_reset_out", None)
        if desired is None and isinstance(out, tuple) and isinstance(out[0], dict) and "desired_goal" in out[0]:
            desired = out[0]["desired_goal"]                    # the env's goal incl. its position noise (base_planner.py:296)
        d_rank, d_world, d_pg = default_shard()
        self.world_size = int(kwargs.get("world_size", d_world))
        self.rank = int(kwargs.get("rank", d_rank if self.world_size == d_world else 0))
        self.process_group = kwargs.get("process_group", d_pg)
        if self.ant_dynamics == "host" and self.world_size > 1:
            raise NotImplementedError("antmaze with a host-side simulator: one rank (the env object is not sharded)")
        self._engine = AntExpansionEngine(
            self.ctx, self.maze, self.start_node.state, self.goal_state, desired_goal=desired, norm=norm,
            edge_length=self.prop_duration_schedule[0], action_horizon=self.action_horizon,
            pred_horizon=getattr(sampler, "pred_horizon", 16), local_map_size=int(lm), local_map_scale=self.local_map_scale,
            s_global=self.s_global, batch=self.batch, capacity=self.capacity, k_steps=getattr(sampler, "num_diffusion_iters", 1),
            early_exit=kwargs.get("early_exit", True), goal_scale=getattr(sampler, "local_map_size", None),
            dynamics=self.ant_dynamics, model=kwargs.get("ant_model"), rank=self.rank, world_size=self.world_size,
            process_group=self.process_group)
        self._engine.ddpm = self._ddpm
        from concurrent.futures import ThreadPoolExecutor
        self._draw_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ditree-draw")

    def _ant_env_step(self, chunk, start, actions, rows):
        """planners/base_planner.py:278-279,290,298 for the alive candidates of one chunk, on the caller's env."""
        A = actions.shape[1]
        out = np.zeros((len(rows), A, 29))
        for k in range(len(rows)):
            self.env.ant_env.set_state(start[k, :15], start[k, 15:])
            for i in range(A):
                out[k, i], _ = self.ant_obs(self.env.step(actions[k, i]))
        return out

    # ------------------------------------------------------------------ reference surface
    def reset(self, start_state=None, goal_state=None, reset_main_path=False):
        if reset_main_path:
            self.init_main_path = None
        if start_state is not None:
            self.start_node = Node(np.asarray(start_state, dtype=np.float64))
            self.goal_state = np.asarray(goal_state, dtype=np.float64)
            to_rc = self.env.maze_data.cell_xy_to_rowcol if self.is_ant else self.env.cell_xy_to_rowcol
            self.options["reset_cell"] = to_rc(start_state[:2])
            if not self.is_ant:
                self.options["reset_deg"] = np.rad2deg(start_state[2])
            self.options["goal_cell"] = to_rc(goal_state[:2])
        self.failed_node_list = []
        self.results = {"iterations": 0, "time": 0, "path": None, "actions": None, "number_of_nodes": 0}
        out = self.env.reset(options=self.options)
        if self.is_ant:
            if isinstance(out, tuple) and isinstance(out[0], dict) and "desired_goal" in out[0]:
                self._engine._desired_arg = np.asarray(out[0]["desired_goal"], dtype=np.float64)[:2].copy()
            self._engine.reset(self.start_node.state, self.goal_state)
            return
        self._engine.reset(self.start_node.state, self.goal_state)
        self._engine.env_goal = np.asarray(self.env.goal, dtype=np.float64)

    def update_maze(self, new_maze):
        self.maze = np.float32(new_maze)
        if not self.is_ant:
            self.env.maze_map = new_maze
        self._engine.update_maze(self.maze)

    def adopt_maze(self, new_maze):
        """Take a known maze whose device copy is already current (online.follow_plan refreshes it in its own
        launch): same bookkeeping as ``update_maze`` without the upload."""
        self.maze = np.float32(new_maze)
        self.env.maze_map = new_maze
        self._engine.maze = self.maze

    @property
    def node_list(self):
        """``Node`` objects of the device tree (RRT.py:42, base_planner.py:24-34).  The tree only grows between two
        ``reset`` calls, so the list is kept and extended by the rows appended since the last access (one D2H of the new
        rows + the visit counters), not rebuilt: scripts that poll ``len(planner.node_list)`` pay O(new nodes)."""
        t = self._engine.tree
        n = t.n_nodes_host
        gen = getattr(self._engine, "generation", 0)
        cache = getattr(self, "_node_cache", None)
        if cache is None or cache[0] != gen or len(cache[1]) > n:
            cache = (gen, [])
        nodes = cache[1]
        k = len(nodes)
        if k < n:
            states = t.state[k:n].cpu().numpy()
            parents = t.parent[k:n].cpu().numpy()
            es, ea = t.edge_states[k:n].cpu().numpy(), t.edge_actions[k:n].cpu().numpy()
            ns, na = t.edge_nstates[k:n].cpu().numpy(), t.edge_nactions[k:n].cpu().numpy()
            for i in range(k, n):
                j = i - k
                if i == 0:
                    nodes.append(Node(states[0]))
                else:
                    nodes.append(Node(states[j], ea[j, : na[j]], es[j, : ns[j]][None], parent=nodes[parents[j]]))
        nv = t.num_visit[:n].cpu().numpy()
        for i in range(n):
            nodes[i].num_visit = int(nv[i])
        self._node_cache = cache
        return nodes

    def nearest_node(self, sample):
        t = self._engine.tree
        q = torch.as_tensor(np.ascontiguousarray(np.asarray(sample, dtype=np.float64)[:, :2]), device=self.ctx.device)
        idx = self.ctx.nn_argmin(q, t.xy, n_nodes=t.n_nodes_host)
        return self.node_list[int(idx[0].item())]          # cached list: O(nodes appended since the last call)

    def check_obstacle_ahead(self, state):
        """RRT.py:61-81 through the device op (the accept kernel evaluates the same function per new node)."""
        self.ctx.upload_maze(np.asarray(self.maze, dtype=np.float32), owner=None)
        st = torch.as_tensor(np.asarray(state, dtype=np.float64).reshape(1, -1), device=self.ctx.device)
        return bool(self.ctx.obstacle_ahead(st)[0].item())

    def extract_path_after_obstacle(self):
        """RRT.py:83-111: xy of ``init_main_path`` from the point nearest to the env state onwards, cut to what lies behind
        the first blocked stretch.  On the device (ditree_path_after_obstacle: nearest point, cell lookup and the two
        searches in one launch on the known maze).  The nearest-point distances are formed in the dtype numpy gives
        ``env.state[:2] - path`` (float64 unless the env state is float32, as right after env.reset), the cells in the
        path's float32."""
        import ctypes as C
        from .._lib import check, lib
        eng = self._engine
        eng.ensure_maze()
        path = np.ascontiguousarray(self.init_main_path, dtype=np.float32)
        p_dev = torch.as_tensor(path, device=self.ctx.device)
        out = torch.zeros(2, dtype=torch.int32, device=self.ctx.device)
        st = np.asarray(self.env.state)
        cur = (C.c_double * 2)(*[float(v) for v in st[:2]])
        check(self.ctx._h, lib().ditree_path_after_obstacle(self.ctx._h, p_dev.data_ptr(), int(path.shape[1]), int(path.shape[0]),
                                                            cur, int(st.dtype == np.float32), out.data_ptr(), self.ctx.stream),
              "path_after_obstacle")
        c, k = (int(v) for v in out.cpu().numpy())
        return self.init_main_path[c:, :2].copy()[k:]

    def draw_round(self, B, remain_init_path=None):
        """B x [sample -> conditioning goal], element for element what the reference's loop draws and leaving both global
        generators where it leaves them -- in bulk (planners/_draw.py: the raw generator output of the whole round is pulled
        once; 8192 candidates in a few ms instead of 76), falling back to the loop itself for anything the bulk path does not
        cover."""
        from ._draw import draw_round_bulk
        out = draw_round_bulk(self, B, remain_init_path)
        return out if out is not None else self._draw_round_loop(B, remain_init_path)

    def _draw_round_loop(self, B, remain_init_path=None):
        """B x [sample -> conditioning goal] in the reference's RNG call order (base_planner.py:162-207,
        RRT.py:134-140,153-156).  run_type 0: random_node_sample, then the goal-conditioning coin.
        run_type > 0: with a remaining reference path, np.random.choice of one of its points kept with
        probability 0.6 (else random_node_sample); the conditioning goal is the sample itself."""
        s = np.zeros((B, self.start_node.state.shape[0]))
        c = np.zeros((B, 2))
        for i in range(B):
            if remain_init_path is not None:
                k = np.random.choice(np.arange(len(remain_init_path)))
                smp = self.random_node_sample() if random.random() < 0.4 else remain_init_path[k][np.newaxis]
            else:
                smp = self.random_node_sample()
            s[i, : smp.shape[1]] = smp[0]
            if self.run_type == 0:
                c[i] = smp[0, :2] if random.random() > self.goal_conditioning_bias else self.goal_state[:2]
            else:
                c[i] = smp[0, :2]
        return s, c

    # ------------------------------------------------------------------ plan
    def _host_actions(self, first, B):
        """A sampler that is not the network (no ``ensure_bound``): the action sequences of a round come from the host
        and by-pass the denoiser (``inject_actions``).  Protocols, in this order:
          ``sampler.sample_round(first_candidate, B, n_chunks, pred_horizon) -> (B, n_chunks, P, 2)``  (tapes: a pure
          function of the global candidate index, so results do not depend on the round size), or a plain callable in
          the reference's shape, ``sampler(obs, prev_actions, goal, local_map)``, called once per (candidate, chunk) in
          candidate-major order with ``obs = prev_actions = local_map = None`` (it cannot see the state: the rounds are
          expanded on the device) and returning (1, P, 2), (P, 2), or one action (1, 2) / (2,) held for the chunk
          -- the shape policies/uniform_policy.py:7-8 returns."""
        eng = self._engine
        if hasattr(self.sampler, "sample_round"):
            a = np.asarray(self.sampler.sample_round(first, B, eng.n_chunks, eng.P), dtype=np.float64)
        else:
            D = eng.ACTION_DIM
            a = np.empty((B, eng.n_chunks, eng.P, D))
            for b in range(B):
                for j in range(eng.n_chunks):
                    v = np.asarray(self.sampler(None, None, self.goal_state[:2], None), dtype=np.float64)
                    a[b, j] = v.reshape(-1, D) if v.size == eng.P * D else v.reshape(1, D)
        if a.shape != (B, eng.n_chunks, eng.P, eng.ACTION_DIM):
            raise ValueError(f"sampler returned actions of shape {a.shape}, need {(B, eng.n_chunks, eng.P, eng.ACTION_DIM)}")
        return torch.as_tensor(np.ascontiguousarray(a), device=self.ctx.device)

    def plan(self):
        eng = self._engine
        dev = self.ctx.device
        network = hasattr(self.sampler, "ensure_bound")
        if network:
            self.sampler.ensure_bound(eng.shard(self.batch)[2])
        start_time = time.time()
        drawn = 0
        goal = None
        has_pm = hasattr(self.env, "prob_map")               # the gym ant env has none (the reference reads it: RRT.py:122)
        orig_prob_map = self.env.prob_map.copy() if has_pm else None
        if self.run_type >= 3:
            self.env.update_prob_map_by_loc()
        remain = None
        if self.run_type > 0 and self.init_main_path is not None:
            remain = self.extract_path_after_obstacle()
        eng.init_main_path = self.init_main_path if self.run_type > 0 else None
        # The samples of round r+1 are drawn (host RNGs, reference call order) by a helper thread while round r runs
        # on the GPU: the C-ABI call releases the GIL, and the draws do not depend on the tree.  The global RNG states
        # are snapshotted before every draw-ahead and restored when the pre-drawn round turns out not to be used (goal
        # reached, budget over), so `random` / `np.random` leave plan() exactly where a run without draw-ahead leaves
        # them.  With batch = 1 nothing is drawn ahead.
        def round_size(already):
            return self.batch if self.max_candidates is None else min(self.batch, self.max_candidates - already)

        ahead = self.batch > 1
        pool = self._draw_pool if ahead else None
        pending = rng_before = None
        cnt = None
        steps_dev = torch.zeros((), dtype=torch.int64, device=dev)     # env steps = two-ball collision tests (cc_calls)
        while eng.agree((time.time() - start_time) < self.time_budget):
            if self.max_candidates is not None and drawn >= self.max_candidates:
                break
            B = round_size(drawn)
            s, c = pending.result() if pending is not None else self.draw_round(B, remain)
            pending = None
            nxt = drawn + B
            if ahead and (self.max_candidates is None or nxt < self.max_candidates):
                rng_before = (random.getstate(), np.random.get_state())
                pending = pool.submit(self.draw_round, round_size(nxt), remain)
            if network:
                # every rank draws the noise of the WHOLE round (same generator state everywhere) and uses its slice
                noise, acts = torch.randn((B, eng.n_chunks, eng.P, eng.ACTION_DIM), device=dev), None
            else:
                noise, acts = None, self._host_actions(drawn, B)
            extra = {}
            if network and eng.ddpm is not None:       # the z of every reverse step, drawn like the start noise (whole round, every rank)
                extra["step_noise"] = torch.randn((B, eng.n_chunks, len(eng.ddpm[0]), eng.P, eng.ACTION_DIM), device=dev)
            if self.is_ant:
                if self.ant_dynamics == "host":
                    extra["step_fn"] = self._ant_env_step
                elif self.ant_dynamics == "tape":
                    extra["next_obs_tape"] = torch.as_tensor(np.ascontiguousarray(self._tape_fn(drawn, B)), device=dev)
            cnt = eng.expand_round(torch.as_tensor(s, device=dev), torch.as_tensor(c, device=dev), noise=noise,
                                   inject_actions=acts, **extra)
            drawn += B
            lo, hi, _ = eng.shard(B)
            steps_dev += eng.rb.chunk_steps[lo:hi].sum()
            goal = int(cnt[CNT_GOAL]) if int(cnt[CNT_GOAL]) >= 0 else None
            if goal is not None:
                break
        if pending is not None:
            pending.result()                     # never leave the helper running on the global RNGs
            random.setstate(rng_before[0])       # the pre-drawn round is dropped: un-draw it
            np.random.set_state(rng_before[1])
        iters = int(cnt[CNT_ITERS]) if cnt is not None else 0
        from ..common import map_utils as _mu
        if not self.is_ant:                                  # is_colliding_ant does not count its calls (map_utils.py:126-136)
            _mu.add_cc_calls(eng.sum_over_ranks(int(steps_dev.item())))   # the counter the drivers read (run_scenarios.py:338,343)
        if has_pm:
            self.env.prob_map = orig_prob_map
        if goal is not None:
            if not self.is_ant:
                self.env.done = True
            return self.handle_goal_reached(goal, iters, start_time)
        node = eng.fallback_node()                          # RRT.py:227-254
        if node is None:
            return self.handle_goal_not_reached(iters, start_time)
        return self.handle_goal_reached(node, iters, start_time)
