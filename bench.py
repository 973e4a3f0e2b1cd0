#!/usr/bin/env python3
"""Candidate tree-expansions/sec (carmaze, H = 32) on N MI355X -- the BASELINE.json metric.

One *step* = one expansion round of the hot path over a batch of synthetic candidates:
nearest node over an N0-node tree snapshot -> 4 chunks x [local map -> conditioning vector ->
denoiser (ResNet-18-GN encoder + FiLM U-Net, one flow step) -> 8 bicycle-model steps with goal
and two-ball collision tests] -> accept/append (plus the RCCL all-gather of candidate records
when N > 1).  Per-GPU batch is fixed (weak scaling); every candidate runs all 4 denoiser calls
(no early-exit compaction), so one candidate = 23.72 GFLOP of algorithmic denoiser work
(SURVEY.md section 8(d)).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision P] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Other workloads of BASELINE.json's config list (each prints its own one-line JSON with `roofline`):
    --workload rollout       config 5: 65 536 car rollouts x T = 16 steps, the rollout kernel alone (no denoiser);
                             --model ant: the higher-DoF rollout slot on the build's stand-in 29-state / 8-action model (NOT MuJoCo)
    --workload mppi          config 5: 65 536 MPPI rollouts (the build's own controller: costs, soft-min update, executed step),
                             sharded over --gpus N with two tiny all-reduces per controller step; --model ant: the same on the
                             stand-in 29-state / 8-action model (config 5 "antmaze" as written; NOT MuJoCo)
    --workload lidar-round   config 4: a round of 8 192 candidates + one 181-ray lidar scan per candidate end pose
    --workload ant-round     config 3: a round of 4 096 ant candidates x H = 48 (24 chunks): ant-sized denoiser + glue + the reference's
                             ant collision / goal tests + accept into the 29-d tree; the env step (MuJoCo in the reference, no oracle) is
                             the build's stand-in model or a tape (--ant-dynamics), labelled in the line
    --workload geometry      the geometry kernels alone (NN, local map, cond vector, rollout chunk, lidar) with GB/s each
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MAC_PER_CALL = 2_939_097_088 + 26_327_808          # U-Net + encoder MACs per denoiser call
PEAK_BF16_TFLOPS = 2500.0                          # dense bf16 MFMA peak, MI355X_MICROARCH.md
H, A, P = 32, 8, 64
N0 = 1024                                          # tree snapshot size


def load_maze(name):
    return np.loadtxt(os.path.join(REPO, "ditreeonlineplanner_amd", "data", f"{name}.csv"), delimiter=",")


def synth_inputs(maze, B_total, seed=20260104):
    """Seeded synthetic round inputs."""
    rng = np.random.default_rng(seed)
    Hh, W = maze.shape
    free = np.argwhere(maze[1:-1, 1:-1] == 0) + 1
    cell = free[rng.integers(0, len(free), N0)]
    # inside the cell with margin 0.25: the two balls (0.075 + 0.1) never reach a neighbour cell
    x = (cell[:, 1] + 0.5) - W / 2 + rng.uniform(-0.25, 0.25, N0)
    y = Hh / 2 - (cell[:, 0] + 0.5) + rng.uniform(-0.25, 0.25, N0)
    nodes = np.stack([x, y, rng.uniform(-np.pi, np.pi, N0), rng.uniform(0, 4, N0), rng.uniform(0, 1, N0),
                      rng.uniform(-0.4, 0.4, N0)], axis=1)
    goal = np.array([(17 + 0.5) - W / 2, Hh / 2 - (2 + 0.5), 0, 0, 0, 0.0])
    # samples as base_planner.py:162-207 (0.15 goal rate), conditioning coin as RRT.py:153-156
    is_goal = rng.random(B_total) <= 0.15
    samples = np.stack([rng.uniform(-W / 2, W / 2, B_total), rng.uniform(-Hh / 2, Hh / 2, B_total),
                        rng.uniform(-np.pi, np.pi, B_total), rng.uniform(-5, 5, B_total),
                        rng.uniform(-1, 1, B_total), rng.uniform(-0.4, 0.4, B_total)], axis=1)
    samples[is_goal] = goal
    coin = rng.random(B_total) > 0.85
    cond = np.where(coin[:, None], samples[:, :2], goal[None, :2])
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(B_total, H // A, P, 2, generator=g)
    return nodes, goal, samples, cond, noise


def cpu_baseline(maze, nodes, goal, samples, cond, noise, state_dict, n_cand=512, batch=64):
    """The oracle (numpy geometry + torch-CPU fp32 denoiser) on a bounded sample of the same workload."""
    from oracle import denoiser as OD
    from oracle import geometry as G
    from oracle import rrt as ORRT
    from oracle import sampler as OS
    # the GPU box gives one GPU's job a 16-core CPU share; more threads than that only oversubscribe
    ncpu = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(ncpu)
    net = OD.init_noise_pred_net().eval()
    net.load_state_dict(state_dict)
    noise_np = noise.numpy()

    def sampler(cand_idx, chunk, state, prev_action, has_prev, cond_goal, local_map):
        cv = OS.car_cond_vector(state, prev_action, has_prev, cond_goal)
        x1 = OS.flow_sample(net, noise_np[cand_idx, chunk], OS.scale_local_map(local_map), cv, k_steps=1)
        return OS.unnormalize_actions(x1)

    pl = ORRT.OraclePlanner(maze, nodes[0], goal, sampler, edge_length=H, action_horizon=A, emulate_sticky_done=False)
    t = pl.tree
    for i in range(1, len(nodes)):
        t.states.append(nodes[i].copy()); t.parents.append(0); t.last_action.append(np.zeros(2))
        t.has_prev.append(True); t.num_visit.append(0); t.edge_states.append(None); t.edge_actions.append(None)
    t0 = time.perf_counter()
    done = 0
    while done < n_cand:
        pl.candidates = done                      # keep the candidate index == noise row
        pl.goal_node = None
        pl.expand_round(samples[done:done + batch], cond[done:done + batch])
        done += batch
        del t.states[N0:], t.parents[N0:], t.last_action[N0:], t.has_prev[N0:], t.num_visit[N0:]
        del t.edge_states[N0:], t.edge_actions[N0:]
    dt = time.perf_counter() - t0
    return dict(value=n_cand / dt, unit="candidate expansions/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n_cand} candidates (rounds of {batch}) of the same synthetic workload, "
                       f"oracle numpy geometry + torch-CPU fp32 denoiser, {dt:.1f} s")


def pmc_traffic(kernel, precision="bf16"):
    """HBM-side bytes per launch of `kernel` (its rocprofv3 name) from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 on
    gfx950 + WRITE_SIZE; profiles/run_profiles.sh + profiles/summarize.py).  Counters cannot be read inside a timed run, so
    this is the RECORDED figure of the same command (same precision), marked as such, or null when no summary is committed."""
    here = os.path.dirname(os.path.abspath(__file__))
    for rnd in ("r04", "r03", "r02"):
        path = os.path.join(here, "profiles", f"{rnd}_{precision}_pmc_traffic.json")
        try:
            with open(path) as f:
                ks = json.load(f)["kernels"]
        except (OSError, KeyError, ValueError):
            continue
        base = kernel.split("<")[0]
        k = ks.get(kernel) or ks.get(base) or next((v for n, v in ks.items() if n.split("<")[0] == base), None)
        if k is None:
            continue
        return {"traffic": k["hbm_bytes_per_launch"], "traffic_kind": "recorded",
                "traffic_unit": "bytes per launch (L2-miss side: FETCH_SIZE*2 + WRITE_SIZE)",
                "traffic_source": os.path.relpath(path, here)}
    return {"traffic": None}


def launch_ranks(argv, n):
    """`python bench.py --gpus N` without a torchrun environment: this process -- which never initialises the GPU --
    starts N fresh rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each),
    forwards rank 0's stdout (the one JSON line), and returns non-zero as soon as any rank does.  Nothing is re-exec'ed
    and no process that has touched the GPU starts another program."""
    import socket
    import subprocess
    rehearse = os.environ.get("DITREE_REHEARSE_ONE_GPU", "0") == "1"
    dry = os.environ.get("DITREE_BENCH_DRYRUN", "0") == "1"
    if not (rehearse or dry):
        have = torch.cuda.device_count()               # counts devices without creating a HIP context on this image
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import signal
    # a terminated launcher must not leave its ranks behind: turn SIGTERM / SIGHUP into an exit that runs the cleanup below
    for sig in (signal.SIGTERM, signal.SIGHUP):
        signal.signal(sig, lambda signum, frame: sys.exit(128 + signum))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DITREE_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    try:
        import threading

        def pump():
            for line in procs[0].stdout:            # the JSON line goes to stdout, library chatter (gloo's banner) to stderr
                text = line.decode(errors="replace")
                dst = sys.stdout if text.lstrip().startswith("{") else sys.stderr
                dst.write(text)
                dst.flush()
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        alive = set(range(n))
        while alive and rc == 0:
            for r in list(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 1
                        print(f"bench.py: rank {r} exited with {code}", file=sys.stderr)
            time.sleep(0.05)
        th.join(timeout=5.0)
    finally:
        for p in procs:                                 # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def recorded_traffic(tag, precision=None, kernel=None):
    """`roofline.traffic` of the side workloads: HBM-side bytes per launch recorded by a separate rocprofv3 --pmc pass of the
    same command (profiles/run_profiles.sh; FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), committed under profiles/.  Counters
    cannot be read inside a timed run, so the field says "recorded" and names its file; null when no pass is committed."""
    base = f"_{tag}" + (f"_{precision}" if precision else "") + "_pmc_traffic.json"
    name = next((r + base for r in ("r04", "r03") if os.path.exists(os.path.join(REPO, "profiles", r + base))), "r04" + base)
    path = os.path.join(REPO, "profiles", name)
    try:
        with open(path) as f:
            ks = json.load(f)["kernels"]
        kname = kernel if kernel else max(ks, key=lambda n: ks[n].get("hbm_bytes_per_launch", 0) * ks[n].get("launches", 1))
        k = ks[kname]
        return {"traffic": k["hbm_bytes_per_launch"], "traffic_kind": "recorded", "traffic_kernel": kname, "traffic_source": "profiles/" + name,
                "traffic_unit": "bytes per launch (L2-miss side: FETCH_SIZE*2 + WRITE_SIZE)"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def _dist_setup(args):
    """-> (rank, world, local, dist module or None, rehearse).  Ranks come from the torchrun-style environment (set by
    `torch.distributed.run` or by launch_ranks above)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal of the N > 1 path on a one-GPU box (tests only, never a measurement): every rank uses device 0 and the
    # candidate records travel through gloo (RCCL refuses two ranks on one device).
    rehearse = os.environ.get("DITREE_REHEARSE_ONE_GPU", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    force_dist = world == 1 and os.environ.get("DITREE_FORCE_DIST", "0") == "1"     # RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_DEBUG", "WARN")          # keep RCCL's banner off stdout: rank 0 prints one JSON line
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    return rank, world, local, dist, rehearse


def _data(rehearse):
    return "synthetic" + (" (one-GPU rehearsal of N ranks over gloo: not a measurement)" if rehearse else "")


def ranks_seen(dist, rehearse, dev):
    """How many ranks the collective backend actually connects: an all-reduce (sum) of ones -- RCCL on the GPUs, gloo in
    the one-GPU rehearsal; 1 without a process group."""
    if dist is None or not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def comm_info(dist, rehearse, dev):
    return {"ranks_seen": ranks_seen(dist, rehearse, dev),
            "backend": "none" if dist is None or not dist.is_initialized() else ("gloo (one-GPU rehearsal)" if rehearse else "nccl (RCCL)")}


def run_dry(args):
    """DITREE_BENCH_DRYRUN=1 (CPU tests of the launcher): rendezvous over gloo, count the ranks, print one line.  No GPU,
    no measurement."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    seen = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(one)
        seen = int(one.item())
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("DITREE_BENCH_DRYRUN_FAIL_RANK", "") == str(rank):
        raise SystemExit(7)
    time.sleep(float(os.environ.get("DITREE_BENCH_DRYRUN_SLEEP", "0")))       # tests of the launcher's cleanup
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "workload": args.workload,
                          "note": "launcher rehearsal on CPU: not a measurement"}), flush=True)


def _timed(step, args, dist, world, dev, rehearse):
    """W warm-up steps, then K steps between barrier + synchronize; the max over ranks (the driver's contract)."""
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        et = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())
    return elapsed


PEAK_HBM_GBS = 8000.0                                # HBM3E peak, MI355X_MICROARCH.md (6.3 TB/s achievable)


def run_rollout(args):
    """BASELINE config 5 -- "65 536 MPPI rollouts" -- as SURVEY.md section 8(d) defines it: the car rollout kernel alone
    (propagate_action_sequence_env + CarEnv.step + is_colliding_car fused), K rollouts of T = 16 steps per launch per GPU,
    every rank its own K (the path shards by rollout, no collective).  Algorithmic bytes per rollout: 48 (state) + 16 T
    (actions, f64) + 48 T (states) + 8 (flags) = 56 + 64 T (the kernel's contract also returns the start row and the (T, 2) copy of
    the executed actions, as the reference does: 1 432 B per rollout).  Rows are stored step-major / candidate-minor in lockstep
    (whole 512-byte runs per wave store)."""
    rank, world, local, dist, rehearse = _dist_setup(args)
    from ditreeonlineplanner_amd.ops import Context
    K = args.batch if args.batch_set else 65536
    T = args.horizon or 16
    maze = load_maze("boxes")
    ctx = Context(local)
    dev = ctx.device
    ctx.upload_maze(maze)
    rng = np.random.default_rng(20260104 + rank)
    Hh, W = maze.shape
    free = np.argwhere(maze[1:-1, 1:-1] == 0) + 1
    cell = free[rng.integers(0, len(free), K)]
    x = (cell[:, 1] + 0.5) - W / 2 + rng.uniform(-0.25, 0.25, K)
    y = Hh / 2 - (cell[:, 0] + 0.5) + rng.uniform(-0.25, 0.25, K)
    st0 = np.stack([x, y, rng.uniform(-np.pi, np.pi, K), rng.uniform(0, 4, K), rng.uniform(0, 1, K),
                    rng.uniform(-0.4, 0.4, K)], axis=1)
    act = np.stack([rng.normal(0.45, 1.0, (K, T)), rng.normal(0.0, 0.92, (K, T))], axis=2)     # the action statistics of carmaze
    s0 = torch.as_tensor(st0, device=dev)
    a = torch.as_tensor(np.ascontiguousarray(act), device=dev)
    state = s0.clone()
    status = torch.zeros(K, dtype=torch.int32, device=dev)
    goal = np.array([(17 + 0.5) - W / 2, Hh / 2 - (2 + 0.5)])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = [0]
    out = {}

    def step():
        state.copy_(s0)
        status.zero_()
        timed = it[0] >= args.warmup
        if timed:
            ev[it[0] - args.warmup][0].record()
        out["r"] = ctx.car_rollout(state, a, goal, A=T, status=status, out=out.get("r"))
        if timed:
            ev[it[0] - args.warmup][1].record()
        it[0] += 1

    elapsed = _timed(step, args, dist, world, dev, rehearse)
    comm = comm_info(dist, rehearse, dev)
    k_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
    if rank == 0:
        st = out["r"][0].cpu().numpy() & 0xFF
        avg_ms = float(np.mean(k_ms))
        alg = K * (56 + 64 * T)
        ach = alg / (avg_ms * 1e-3) / 1e9
        res = {"metric": "car rollouts/sec (T=16 bicycle steps + goal + two-ball collision per step, no denoiser)",
               "value": K * world * args.steps / elapsed, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": _data(rehearse),
               "config": {"workload": f"BASELINE config 5 as SURVEY 8(d) defines it: {K} car rollouts x T={T} per GPU on boxes.csv, "
                                      "rollout kernel alone (the reference has no MPPI module and no ant MPPI script)",
                          "rollouts_per_gpu": K, "horizon": T, "parallelism": f"rollouts sharded x{world}, no collective"},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                            **recorded_traffic("rollout", kernel="car_rollout_kernel"), "kernel": "car_rollout_kernel", "avg_launch_ms": avg_ms,
                            "algorithmic_bytes_per_launch": alg,
                            "note": "bound by its sequential FP64 chain at one wave per SIMD (2 sincos + cos + tanh + 2 hypot + sqrt per step), not by HBM: the launch without row stores takes 80 % of this one (DESIGN.md section 5)"},
               "outcome": {"ok": int((st == 0).sum()), "goal": int((st == 1).sum()), "collided": int((st == 2).sum())}, **comm}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_rollout_ant(args):
    """The higher-DoF dynamics slot (BASELINE configs 3 / 5 "antmaze") measured alone: K rollouts of
    T = 16 env steps of the build's STAND-IN 29-state / 8-action crawler model (frame_skip 5 sub-steps each; NOT MuJoCo, parity
    unpinned), every step followed by the reference's ant goal test and is_colliding_ant, rows stored step-major / candidate-
    minor.  Algorithmic bytes per rollout: 232 + 64 T (actions) + 232 T (states) + 8."""
    rank, world, local, dist, rehearse = _dist_setup(args)
    from ditreeonlineplanner_amd.ops import Context
    K = args.batch if args.batch_set else 65536
    T = args.horizon or 16
    maze = load_maze("boxes")
    ctx = Context(local)
    dev = ctx.device
    ctx.upload_maze(maze)
    nodes, *_ = ant_synth(maze, K, 1, 1, 20260104 + rank)
    rng = np.random.default_rng(99 + rank)
    act = np.clip(rng.uniform(-1, 1, (K, 1, 8)) + rng.normal(0, 0.4, (K, T, 8)), -1.2, 1.2)
    s0 = torch.as_tensor(nodes, device=dev)
    a = torch.as_tensor(np.ascontiguousarray(act), device=dev)
    state = s0.clone()
    status = torch.zeros(K, dtype=torch.int32, device=dev)
    Hh, W = maze.shape
    goal = np.array([((17 + 0.5) - W / 2) * 4.0, (Hh / 2 - (2 + 0.5)) * 4.0])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = [0]
    out = {}
    import ctypes as C
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd._lib import check, lib
    from ditreeonlineplanner_amd.ops import _dbl, _ptr
    states = torch.zeros(T + 1, 29, K, dtype=torch.float64, device=dev).permute(2, 0, 1)
    aout = torch.zeros(T, 8, K, dtype=torch.float64, device=dev).permute(2, 0, 1)
    steps_t = torch.zeros(K, dtype=torch.int32, device=dev)
    model = _lib.AntModel.default()
    sl, al = _lib.Strides(*states.stride()), _lib.Strides(*aout.stride())
    g, gp = _dbl(goal)

    def step():
        state.copy_(s0)
        status.zero_()
        timed = it[0] >= args.warmup
        if timed:
            ev[it[0] - args.warmup][0].record()
        check(ctx._h, lib().ditree_ant_rollout(ctx._h, C.byref(model), _ptr(state), _ptr(a), T * 8, None, 0, _ptr(status), K, T, gp,
                                               0.45 * 4.0, 1.2, 4.0, _ptr(states), C.byref(sl), _ptr(aout), C.byref(al), _ptr(steps_t),
                                               ctx.stream), "ant_rollout")
        if timed:
            ev[it[0] - args.warmup][1].record()
        it[0] += 1

    elapsed = _timed(step, args, dist, world, dev, rehearse)
    comm = comm_info(dist, rehearse, dev)
    k_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
    if rank == 0:
        st = status.cpu().numpy() & 0xFF
        executed = int(steps_t.sum().item())
        avg_ms = float(np.mean(k_ms))
        alg = K * (232 + 64 * T + 232 * T + 8)
        ach = alg / (avg_ms * 1e-3) / 1e9
        res = {"metric": "ant-model rollouts/sec (T=16 env steps x 5 sub-steps of the build's stand-in 29-state / 8-action crawler model, NOT MuJoCo; "
                         "goal test + is_colliding_ant per step; no denoiser)",
               "value": K * world * args.steps / elapsed, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": _data(rehearse) + "; dynamics: stand-in model, parity unpinned, not MuJoCo",
               "config": {"workload": f"BASELINE config 5's antmaze variant / SURVEY 8(f)4: {K} rollouts x T={T} of the higher-DoF rollout kernel per GPU on "
                                      "boxes.csv x 4 (stand-in dynamics; the reference has neither an ant MPPI script nor the ant physics in its repository)",
                          "rollouts_per_gpu": K, "horizon": T, "frame_skip": 5, "parallelism": f"rollouts sharded x{world}, no collective"},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                            **recorded_traffic("rollout_ant", kernel="ant_rollout_kernel"), "kernel": "ant_rollout_kernel<true>",
                            "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg,
                            "executed_env_steps_per_launch": executed,
                            "note": "rows stored step-major / candidate-minor (512 contiguous bytes per wave store); the kernel is bound by its "
                                    "sequential FP64 chain with one wave per SIMD, see DESIGN.md"},
               "outcome": {"ok": int((st == 0).sum()), "goal": int((st == 1).sum()), "collided": int((st == 2).sum())}, **comm}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_mppi(args):
    """BASELINE config 5 -- "65 536 MPPI rollouts sharded over 8 x MI355X, pure-rollout stress, no diffusion" -- with the
    build's own MPPI controller (the reference ships none: PARITY UNPINNED, include/ditree.h ditree_mppi_step).  One step =
    one controller step: K rollouts x T = 16 [car dynamics, two-ball collision, goal test, nearest reference-path point],
    costs, soft-min weights, weighted control update, one executed env step.  --gpus N shards the K GLOBAL rollouts over the
    ranks (two all-reduces of 1 and 3 + 2T doubles per step: strong scaling); default K = 65 536 per node.
    Roofline of the dominant kernel (mppi_rollout_kernel): it is an FP64 chain with on-device noise -- algorithmic HBM bytes
    are the K costs + flags it writes (12 B per rollout) -- so the `hbm` fraction is tiny by design; the governing unit is the
    FP64 vector pipe, reported next to it as `fp64`."""
    rank, world, local, dist, rehearse = _dist_setup(args)
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.mppi import MPPI
    from ditreeonlineplanner_amd.ops import Context
    # --global-batch K: K rollouts split over the ranks (strong scaling; default 65 536 = BASELINE config 5);
    # --batch K: K rollouts PER GPU (weak scaling: the step is latency-bound at 8 192 rollouts per GPU, so this is the mode
    # in which more GPUs buy more samples per controller step)
    weak = args.batch_set and not args.global_batch
    K = args.batch * world if weak else (args.global_batch or 65536)
    T = args.horizon or 16
    maze = load_maze("boxes")
    ctx = Context(local)
    dev = ctx.device
    Hh, W = maze.shape
    xy = lambda r, c: np.array([(c + 0.5) - W / 2, Hh / 2 - (r + 0.5)])              # noqa: E731
    a_, b_, c_ = xy(18, 1), xy(18, 18), xy(1, 18)
    seg1 = a_ + (b_ - a_) * np.linspace(0, 1, 850)[:, None]
    seg2 = b_ + (c_ - b_) * np.linspace(0, 1, 850)[1:, None]
    path = np.concatenate([seg1, seg2])
    ant = args.model == "ant"
    if ant:
        # config 5 as written ("MPPI antmaze"): the same controller on the higher-DoF slot -- the build's stand-in crawler model
        # (NOT MuJoCo), the reference's ant collision / goal tests, boxes.csv scaled by s_global = 4
        path = path * 4.0
        m = MPPI(maze_data=maze, T=T, K=K, nx=29, nu=8, seed=20260104, ctx=ctx, rank=rank, world_size=world)
        start = np.zeros(29)
        start[:2] = path[0]
        start[2], start[3] = 0.75, 1.0
        start[7:15] = np.tile([0.0, 0.87], 4)
        goal_s = np.zeros(29)
        goal_s[:2] = c_ * 4.0
        m.reset(start_state=start, goal_state=goal_s)
    else:
        m = MPPI(maze_data=maze, T=T, K=K, nx=6, nu=2, seed=20260104, ctx=ctx, rank=rank, world_size=world, lanes=args.mppi_lanes)
        start = np.array([path[0, 0], path[0, 1], 0.0, 0.0, 0.0, 0.0])
        m.reset(start_state=start, goal_state=np.array([c_[0], c_[1], 0, 0, 0, 0.0]))
    m.set_ref_path(path)
    m._state.copy_(torch.as_tensor(start))              # the state then stays on the device (the step writes it back)
    state = [start]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = [0]

    def allred(t, op):                                   # RCCL on the device; through the host in the one-GPU rehearsal (gloo)
        if rehearse:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def step():
        # the driver's loop: step() returns the next state to the host (one 64-byte D2H + sync per controller step)
        timed = it[0] >= args.warmup
        if timed:
            ev[it[0] - args.warmup][0].record()
        m.launch(_lib.MPPI_ROLLOUTS)                       # the dominant kernel, bracketed on its own
        if timed:
            ev[it[0] - args.warmup][1].record()
        if world > 1:
            m.launch(_lib.MPPI_MIN)
            allred(m._result[3:4], dist.ReduceOp.MIN)
            m.launch(_lib.MPPI_SUMS)
            allred(m._sums, dist.ReduceOp.SUM)
            m.launch(_lib.MPPI_APPLY | _lib.MPPI_EXECUTE)
        else:
            m.launch(_lib.MPPI_MIN | _lib.MPPI_SUMS | _lib.MPPI_APPLY | _lib.MPPI_EXECUTE)
        m.counter += 1
        res = m._result.cpu().numpy()              # one D2H + sync per step: action, status, statistics, the new state
        state[0] = res[24:53].copy() if ant else res[8:14].copy()
        it[0] += 1

    elapsed = _timed(step, args, dist, world, dev, rehearse)
    comm = comm_info(dist, rehearse, dev)
    if rank == 0:
        k_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
        Kloc = m.K_local
        alg = Kloc * 12 + (232 + 64 * T if ant else 48 + 32 * T) + len(path) * 16
        ach = alg / (k_ms * 1e-3) / 1e9
        # FP64 work of a rollout step (counted from the kernel source, fma = 2), an UPPER BOUND on the executed work: rollouts that
        # hit a wall or the goal stop early (`controller.collided_rollouts_last_step`).  car: dynamics ~60 + 3 sin/cos pairs + tanh
        # (~40 each) + two-ball collision ~2 x 60 + goal 6 + path window 64 points x 7 + noise hash / Box-Muller ~120 -> ~900 flop;
        # ant stand-in: 5 sub-steps x (4 legs x ~40 + tanh x 4 + atan2 + sin / cos + quaternion ~60 = ~420) + collision ~70 +
        # path window 64 x 7 + 4 noise pairs ~480 -> ~3100 flop
        fps = 3100.0 if ant else 900.0
        flop = Kloc * T * fps
        kname = "mppi_ant_rollout_kernel" if ant else f"mppi_rollout_kernel<{args.mppi_lanes or 2}>"
        what = ("ant-slot rollouts: the build's stand-in 29-state / 8-action crawler model, NOT MuJoCo" if ant else "car rollouts")
        res = {"metric": f"MPPI rollouts/sec (controller step: K x T=16 {what} with collision / goal / path-tracking cost, soft-min update, one executed step)",
               "value": K * args.steps / elapsed, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
               "dtype": "f64", "data": _data(rehearse),
               "config": {"workload": f"BASELINE config 5: {K} MPPI rollouts (global) x T={T}, " +
                                      ("the stand-in ant model (NOT MuJoCo; parity unpinned) on boxes.csv x 4" if ant else "car model on boxes.csv") +
                                      f", L-shaped reference path of {len(path)} points, on-device noise; the build's own controller (the reference ships no MPPI module: parity unpinned)",
                          "rollouts_global": K, "rollouts_per_gpu": Kloc, "horizon": T,
                          "parallelism": f"rollouts sharded x{world}; all-reduce MIN (1 double) + SUM ({3 + 2 * T} doubles) per step"},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                            **recorded_traffic("mppi_ant" if ant else "mppi", kernel="mppi_ant_rollout_kernel" if ant else "mppi_rollout_kernel"), "kernel": kname,
                            "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": alg, "kernel_time_share": k_ms * 1e-3 * args.steps / elapsed,
                            "fp64": {"achieved_tflops": flop / (k_ms * 1e-3) / 1e12, "peak_tflops": 78.6,
                                     "frac": flop / (k_ms * 1e-3) / 1e12 / 78.6, "flop_per_rollout_step": fps,
                                     "note": "the governing unit: a sequential FP64 chain per rollout" + ("" if ant else f" ({args.mppi_lanes or 2} lanes share one rollout)") +
                                             "; flop count from the source, fma = 2; UPPER BOUND on executed work (no early termination assumed)"},
                            "note": "noise is generated on the device: the kernel writes 12 B per rollout and reads ~30 KB of shared inputs per work-group out of L2"},
               "controller": {"state_xy": [float(state[0][0]), float(state[0][1])], **m.last}, **comm}
        res["controller"].update({"collided_rollouts_last_step": int(m._result[6].item()), "effective_samples_last_step": float(m._result[7].item())})
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_geometry(args):
    """The geometry kernels of the path measured alone (HBM-bound nominally): nearest node, local map,
    conditioning vector, lidar scan, rollout chunk (A = 8), accept -- at the config-4 round size (8192 candidates, 65 536-node
    tree for the NN scan).  One JSON line; `roofline` is the rollout chunk's (the largest of them inside a round),
    `kernels` lists every kernel's algorithmic bytes, time and GB/s."""
    rank, world, local, dist, rehearse = _dist_setup(args)
    from ditreeonlineplanner_amd.ops import Context
    B = args.batch if args.batch_set else 8192
    N = 65536
    maze = load_maze("boxes")
    ctx = Context(local)
    dev = ctx.device
    ctx.upload_maze(maze)
    rng = np.random.default_rng(20260104 + rank)
    Hh, W = maze.shape
    free = np.argwhere(maze[1:-1, 1:-1] == 0) + 1

    def poses(n):
        cell = free[rng.integers(0, len(free), n)]
        x = (cell[:, 1] + 0.5) - W / 2 + rng.uniform(-0.25, 0.25, n)
        y = Hh / 2 - (cell[:, 0] + 0.5) + rng.uniform(-0.25, 0.25, n)
        return np.stack([x, y, rng.uniform(-np.pi, np.pi, n), rng.uniform(0, 4, n), rng.uniform(0, 1, n),
                         rng.uniform(-0.4, 0.4, n)], axis=1)
    nodes = torch.as_tensor(poses(N), device=dev)
    node_xy = nodes[:, :2].contiguous()
    node_la = torch.zeros(N, 2, dtype=torch.float64, device=dev)
    node_hp = torch.ones(N, dtype=torch.uint8, device=dev)
    st = torch.as_tensor(poses(B), device=dev)
    q = torch.as_tensor(np.stack([rng.uniform(-W / 2, W / 2, B), rng.uniform(-Hh / 2, Hh / 2, B)], axis=1), device=dev)
    prev = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    hasp = torch.ones(B, dtype=torch.uint8, device=dev)
    goal = torch.as_tensor(np.tile(np.array([7.5, 7.5]), (B, 1)), device=dev)
    acts = torch.as_tensor(np.stack([rng.normal(0.45, 1.0, (B, 8)), rng.normal(0.0, 0.92, (B, 8))], axis=2).copy(), device=dev)
    cell_pose = torch.stack([st[:, 0] + W / 2, Hh / 2 - st[:, 1], st[:, 2]], dim=1).contiguous()
    true_maze = torch.as_tensor(maze.astype(np.float32), device=dev)
    lm = torch.empty(B, 20, 20, dtype=torch.float32, device=dev)
    state = st.clone()
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    keep = {}

    def k_rollout():
        state.copy_(st)
        status.zero_()
        keep["r"] = ctx.car_rollout(state, acts, np.array([7.5, 7.5]), A=8, status=status, out=keep.get("r"))
    kernels = {
        "nn_argmin_kernel": (lambda: ctx.nn_argmin(q, node_xy, gather=(nodes, node_la, node_hp)), N * 16 + B * (16 + 4 + 72)),
        "local_map_kernel": (lambda: ctx.local_map(st, n=20, scale=0.2, s_global=1.0, scaled=True, out=lm), B * (24 + 400 * 4)),
        "cond_vector_kernel": (lambda: ctx.cond_vector(st, prev, hasp, goal), B * 109),
        "car_rollout_kernel (A = 8)": (k_rollout, B * (56 + 64 * 8)),
        "lidar_scan_kernel": (lambda: ctx.lidar_scan(cell_pose, true_maze, want_visited=True), B * (24 + 181 * 25 + maze.size) + maze.size * 4),
    }
    res_k = {}
    for name, (fn, alg) in kernels.items():
        for _ in range(args.warmup):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        res_k[name] = {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / (ms * 1e-3) / 1e9,
                       "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
    comm = comm_info(dist, rehearse, dev)
    if rank == 0:
        total_ms = sum(v["avg_launch_ms"] for v in res_k.values())
        ro = res_k["car_rollout_kernel (A = 8)"]
        res = {"metric": "geometry kernels of one chunk at the config-4 round size (NN + local map + cond + 8-step rollout + lidar)",
               "value": B / (total_ms * 1e-3), "unit": "candidates/s through the geometry kernels alone", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": total_ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": _data(rehearse),
               "config": {"workload": f"geometry kernels alone: {B} candidates, {N}-node tree, boxes.csv; includes per-call tensor "
                                      "allocation of the Python front end (event-timed around the front-end call)"},
               "roofline": {"bound": "hbm", "achieved": ro["achieved_GBps"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": ro["frac_of_hbm_peak"], **recorded_traffic("geometry", kernel="car_rollout_kernel"), "kernel": "car_rollout_kernel (A = 8)",
                            "avg_launch_ms": ro["avg_launch_ms"], "algorithmic_bytes_per_launch": ro["algorithmic_bytes_per_launch"]},
               "kernels": res_k, **comm}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_lidar_round(args):
    """BASELINE config 4: the car round with the lidar in the loop -- a round of B candidates (B = 8192 global; NN ->
    4 x [map, cond, denoiser, 8 steps] -> accept) followed by one 181-ray `Lidar2DSim.scan` per candidate end pose on
    the true maze (an extrapolation, the reference scans one pose every 0.2 s)."""
    rank, world, local, dist, rehearse = _dist_setup(args)
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.engine import CNT_GOAL, CNT_LATCH, CNT_NODES, ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context
    Bglob = args.global_batch or (args.batch * world if args.batch_set else 8192)
    if Bglob % world:
        raise SystemExit("--global-batch must divide by the number of GPUs")
    Bper = Bglob // world
    maze = load_maze("boxes")
    nodes, goal, samples, cond, noise = synth_inputs(maze, Bglob)
    ctx = Context(local)
    dev = ctx.device
    net = NoisePredNet(seed=0)
    net.bind(ctx, precision=_lib.PREC_NAMES[args.precision], max_batch=Bper)
    eng = ExpansionEngine(ctx, maze, nodes[0], goal, edge_length=H, action_horizon=A, pred_horizon=P, batch=Bglob,
                          capacity=N0 + Bglob, rank=rank, world_size=world, emulate_sticky_done=False)
    t = eng.tree
    nd = torch.as_tensor(nodes, device=dev)
    t.state[:N0] = nd
    t.xy[:N0] = nd[:, :2]
    t.parent[:N0] = torch.arange(-1, N0 - 1, device=dev, dtype=torch.int32).clamp(min=0)
    t.parent[0] = -1
    t.has_prev[1:N0] = 1          # the root has no previous action (as the oracle tree)

    def reset_tree():
        t.counters[CNT_NODES] = N0
        t.counters[CNT_GOAL] = -1
        t.counters[CNT_LATCH] = 0
        t.n_nodes_host = N0
    reset_tree()
    s_dev = torch.as_tensor(samples, device=dev)
    c_dev = torch.as_tensor(cond, device=dev)
    n_dev = noise.to(dev)
    true_maze = torch.as_tensor(maze.astype(np.float32), device=dev)
    lo, hi, _ = eng.shard(Bglob)
    Hh, W = maze.shape
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    it = [0]
    keep = {}

    def step():
        eng.expand_round(s_dev, c_dev, noise=n_dev)
        reset_tree()
        # pose in cell units (x_col, y_row, yaw) of this rank's candidates' end states (scan_and_update_maze, :112-127)
        es = eng.rb.end_state[lo:hi]
        poses = torch.stack([es[:, 0] + W / 2, Hh / 2 - es[:, 1], es[:, 2]], dim=1).contiguous()
        timed = it[0] >= args.warmup
        if timed:
            ev[it[0] - args.warmup][0].record()
        keep["scan"] = ctx.lidar_scan(poses, true_maze, want_visited=True)
        if timed:
            ev[it[0] - args.warmup][1].record()
        it[0] += 1

    if world > 1:
        eng.exchange_events = []
    elapsed = _timed(step, args, dist, world, dev, rehearse)
    comm = comm_info(dist, rehearse, dev)
    if eng.exchange_events:
        ex = eng.exchange_events[-args.steps:]
        comm["exchange"] = {"ms_per_round": float(np.mean([a.elapsed_time(b) for a, b in ex])), "bytes_per_rank": Bper * 96}
    if rank == 0:
        avg_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
        alg = Bper * (24 + 181 * 25) + maze.size * 4 + Bper * maze.size          # poses + per-ray outputs + maze + visited bitmap
        ach = alg / (avg_ms * 1e-3) / 1e9
        res = {"metric": "candidate tree-expansions/sec (carmaze, H=32, lidar scan per candidate end pose)",
               "value": Bglob * args.steps / elapsed, "unit": "candidate expansions/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": _data(rehearse),
               "config": {"workload": f"BASELINE config 4: carmaze round of {Bglob} candidates (global) + one 181-ray lidar scan per "
                                      f"candidate end pose, boxes.csv, {N0}-node snapshot, seeded random weights",
                          "global_batch": Bglob, "batch_per_gpu": Bper, "parallelism": f"candidates sharded x{world}"},
               "roofline": {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                            **recorded_traffic("lidar_round", kernel="lidar_scan_kernel"), "kernel": "lidar_scan_kernel", "avg_launch_ms": avg_ms,
                            "algorithmic_bytes_per_launch": alg, "kernel_time_share": avg_ms * 1e-3 * args.steps / elapsed,
                            "note": "the round itself is MFMA-bound (see the default workload); this is the lidar kernel's line"},
               **comm}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def ant_synth(maze, N0, B, n_chunks, seed):
    """Seeded synthetic snapshot + round inputs of BASELINE config 3: N0 nodes in free cells of the maze scaled by s_global = 4
    (inside the cell with a margin the 1.2 ball never leaves), upright torsos, joints at rest, 3-row histories; samples as
    planners/base_planner.py:193-207 (0.15 goal rate), conditioning coin as planners/RRT.py:153-156."""
    rng = np.random.default_rng(seed)
    Hh, W = maze.shape
    free = np.argwhere(maze[1:-1, 1:-1] == 0) + 1
    cell = free[rng.integers(0, len(free), N0)]
    nodes = np.zeros((N0, 29))
    nodes[:, 0] = ((cell[:, 1] + 0.5) - W / 2) * 4.0 + rng.uniform(-0.6, 0.6, N0)
    nodes[:, 1] = (Hh / 2 - (cell[:, 0] + 0.5)) * 4.0 + rng.uniform(-0.6, 0.6, N0)
    nodes[:, 2] = rng.uniform(0.5, 0.8, N0)
    q = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.05, (N0, 4))
    nodes[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    nodes[:, 7:15] = np.tile([0.0, 0.87], 4) + rng.normal(0, 0.05, (N0, 8))
    nodes[:, 15:] = rng.normal(0, 0.3, (N0, 14))
    hist = np.repeat(nodes[:, None, :], 3, axis=1)
    hist[:, :2, 7:] += rng.normal(0, 0.03, (N0, 2, 22))
    goal = np.zeros(29)
    goal[:2] = [((17 + 0.5) - W / 2) * 4.0, (Hh / 2 - (2 + 0.5)) * 4.0]
    is_goal = rng.random(B) <= 0.15
    samples = np.zeros((B, 29))
    samples[:, 0] = rng.uniform(-4.0 * W / 2, 4.0 * W / 2, B)
    samples[:, 1] = rng.uniform(-4.0 * Hh / 2, 4.0 * Hh / 2, B)
    samples[is_goal] = goal
    coin = rng.random(B) > 0.85
    cond = np.where(coin[:, None], samples[:, :2], goal[None, :2])
    noise = torch.randn(B, n_chunks, 16, 8, generator=torch.Generator().manual_seed(seed))
    last_action = rng.uniform(-1, 1, (N0, 8))
    return nodes, hist, last_action, goal, samples, cond, noise


def ant_cpu_baseline(maze, nodes, hist, last_action, goal, samples, cond, noise, state_dict, norm, n_cand=128, batch=64):
    """The oracle (kind "port": numpy f64 glue incl. the reference-pinned collision / goal tests and the stand-in model, torch-CPU
    fp32 denoiser) on a bounded sample of the same round."""
    from oracle import ant as OA
    from oracle import denoiser as OD
    from oracle import sampler as OS
    n_thr = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(n_thr)
    net = OD.init_noise_pred_net(input_dim=8, action_dim=8, obs_dim=29, obs_history=3, action_history=1).eval()
    net.load_state_dict(state_dict)
    nz = noise.numpy()
    meta = dict(Observations_mean=norm[:27], Observations_std=norm[27:54], Actions_mean=norm[54:62], Actions_std=norm[62:70])

    def sampler(cand_idx, chunk, h, prev_a, has_prev, cond_goal, lm):
        cv = OS.ant_cond_vector(h, prev_a, has_prev, cond_goal, meta=meta)
        x = OS.flow_sample(net, nz[cand_idx, chunk], OS.scale_local_map(lm), cv, k_steps=1)
        return x.astype(np.float64) * meta["Actions_std"] + meta["Actions_mean"]
    pl = OA.OracleAntPlanner(maze, nodes[0], goal, goal[:2], sampler, lambda c, j, i, cur, act: OA.ant_model_step(cur, act))
    N0 = len(nodes)
    pl.states = [s for s in nodes]
    pl.parents = [-1] + list(range(0, N0 - 1))
    pl.last_action = [np.zeros(8)] + [a for a in last_action[1:]]
    pl.has_prev = [False] + [True] * (N0 - 1)
    pl.edge_states = [None] + [h for h in hist[1:]]
    pl.edge_actions = [None] * N0
    t0 = time.perf_counter()
    done = 0
    while done < n_cand:
        pl.candidates = done
        pl.goal_node = None
        pl.expand_round(samples[done:done + batch], cond[done:done + batch])
        del pl.states[N0:], pl.parents[N0:], pl.last_action[N0:], pl.has_prev[N0:], pl.edge_states[N0:], pl.edge_actions[N0:]
        done += batch
    dt = time.perf_counter() - t0
    return {"value": n_cand / dt, "unit": "candidate expansions/s", "cores": n_thr, "kind": "port",
            "sample": f"{n_cand} candidates of the same round in rounds of {batch} (oracle: numpy f64 glue + stand-in model, torch-CPU fp32 "
                      f"denoiser, {n_thr} threads; early exit as the reference abandons collided edges)", "seconds": dt}


def run_ant_round(args):
    """BASELINE config 3 (cfgs/antmaze.yaml + fm_policy, B = 4096 candidates, H = 48) as a real expansion round: nearest node
    over a 1024-node snapshot -> 24 chunks (action_horizon 2) x [16 x 16 @ 0.8 local map (s_global 4), ant conditioning vector
    incl. quaternion -> rot6d and the 3-step history, ResNet-18-GN encoder on 16 x 16, FiLM U-Net at input_dim 8 / pred_horizon
    16 / cond 497, flow step, un-normalise, 2 env steps each followed by the reference's goal test and is_colliding_ant] ->
    accept into the 29-d tree.  24 x 1.536 GFLOP = 36.86 GFLOP per candidate when every chunk runs.
    THE ENV STEP is the build's stand-in crawler model (`--ant-dynamics model`, default) or a next-observation tape (`tape`):
    MuJoCo (the reference's physics) has no oracle here and is NOT built -- said so in `metric`, `data` and `config`.  The headline
    `value` runs every chunk of every candidate (no early exit, as the car headline); `early_exit` reports the compacted rate."""
    rank, world, local, dist, rehearse = _dist_setup(args)
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.engine import CNT_GOAL, CNT_NODES, AntExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context
    Bper = args.batch if args.batch_set else 4096
    Btot = Bper * world
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must divide by the number of GPUs")
        Btot = args.global_batch
        Bper = Btot // world
    nC, A_ant = 48 // 2, 2
    prec = args.precision
    dyn = args.ant_dynamics
    maze = load_maze("boxes")
    ctx = Context(local)
    dev = ctx.device
    net = NoisePredNet(input_dim=8, additional_global_cond_dim=97, pred_horizon=16, local_map_size=16, seed=0)
    net.bind(ctx, precision=_lib.PREC_NAMES[prec], max_batch=Bper)
    with open(os.path.join(REPO, "ditreeonlineplanner_amd", "data", "metadata_antmaze.json")) as f:
        md = json.load(f)
    norm = np.array(md["Observations_mean"] + md["Observations_std"] + md["Actions_mean"] + md["Actions_std"])
    nodes, hist, last_action, goal, samples, cond, noise = ant_synth(maze, N0, Btot, nC, 20260104)
    force_dist = world == 1 and dist is not None

    def make_engine(early_exit):
        eng = AntExpansionEngine(ctx, maze, nodes[0], goal, norm=norm, batch=Btot, capacity=N0 + Btot, dynamics=dyn,
                                 early_exit=early_exit, rank=rank, world_size=world)
        eng.force_allgather = force_dist
        t = eng.tree
        nd = torch.as_tensor(nodes, device=dev)
        t.state[:N0] = nd
        t.xy[:N0] = nd[:, :2]
        t.parent[:N0] = torch.arange(-1, N0 - 1, device=dev, dtype=torch.int32).clamp(min=0)
        t.parent[0] = -1
        t.has_prev[1:N0] = 1
        t.last_action[1:N0] = torch.as_tensor(last_action[1:], device=dev)
        t.hist[:N0] = torch.as_tensor(hist, device=dev)
        t.hist_n[:N0] = 3
        t.hist_n[0] = 1
        return eng
    s_dev, c_dev, n_dev = torch.as_tensor(samples, device=dev), torch.as_tensor(cond, device=dev), noise.to(dev)
    tape = None
    if dyn == "tape":
        rng = np.random.default_rng(7)
        tp = np.zeros((Btot, nC * A_ant, 29))
        base = nodes[rng.integers(0, N0, Btot)]
        tp[:] = base[:, None, :]
        tp[:, :, :2] += np.cumsum(rng.normal(0, 0.05, (Btot, nC * A_ant, 2)), axis=1)
        tape = torch.as_tensor(tp.reshape(Btot, nC, A_ant, 29), device=dev)
    comm = comm_info(dist, rehearse, dev)

    def run(eng):
        t = eng.tree

        def step():
            eng.expand_round(s_dev, c_dev, noise=n_dev, next_obs_tape=tape)
            t.counters[CNT_NODES] = N0
            t.counters[CNT_GOAL] = -1
            t.n_nodes_host = N0
        return step
    eng = make_engine(False)
    step = run(eng)
    for _ in range(args.warmup):
        step()
    ctx.profile(1)
    if world > 1 or force_dist:
        eng.exchange_events = []
    elapsed = _timed(step, dataclass_replace(args, warmup=0), dist, world, dev, rehearse)
    prof = ctx.profile_read(prec)
    ctx.profile(0)
    ctx.check_range()
    if eng.exchange_events:
        comm["exchange"] = {"ms_per_round": float(np.mean([a.elapsed_time(b) for a, b in eng.exchange_events])),
                            "bytes_per_rank": Bper * eng.tree.record_doubles * 8}
    eng.exchange_events = None
    lo, hi, _ = eng.shard(Btot)
    st = eng.rb.status[lo:hi].cpu().numpy() & 0xFF
    run_chunks = eng.rb.chunks_run[lo:hi].cpu().numpy()
    ee = None
    if not args.no_early_exit_line:
        eng2 = make_engine(True)
        step2 = run(eng2)
        for _ in range(max(1, args.warmup)):
            step2()
        n2 = max(2, args.steps // 2)
        el2 = _timed(step2, dataclass_replace(args, warmup=0, steps=n2), dist, world, dev, rehearse)
        ee = {"value": Btot * n2 / el2, "ms_per_step": 1e3 * el2 / n2, "denoiser_calls_per_candidate": float(run_chunks.mean()),
              "note": "later chunks run on the still-alive candidates only (planners/RRT.py:179-184); same tree, bit for bit"}
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        cpu = ant_cpu_baseline(maze, nodes, hist, last_action, goal, samples, cond, noise, sd, norm)
    if rank == 0:
        mac = 752_250_880 + 15_749_120
        alg = 2.0 * mac * Bper * nC * args.steps
        all_ms = sum(v["ms"] for v in prof.values())
        peak = 157.3 if prec == "f32" else (PEAK_BF16_TFLOPS / 3.0 if prec in ("f16x3", "bf16x3") else PEAK_BF16_TFLOPS)
        ach = alg / (all_ms * 1e-3) / 1e12
        stand_in = ("the build's stand-in crawler model (NOT MuJoCo, parity unpinned)" if dyn == "model"
                    else "a synthetic next-observation tape (NOT MuJoCo)")
        res = {"metric": f"candidate tree-expansions/sec (antmaze, H=48; env step = {stand_in})",
               "value": Btot * args.steps / elapsed, "unit": "candidate expansions/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": prec,
               "data": _data(rehearse) + f"; env step: {stand_in}",
               "config": {"workload": f"BASELINE config 3: cfgs/antmaze.yaml + fm_policy, batch={Bper} candidates per GPU, H=48 = 24 chunks x "
                                      "[local map 16x16@0.8 s_global 4, ant cond vector (rot6d, 3-step history), encoder + U-Net P=16 D=8 cond 497, "
                                      "flow step, 2 of 16 actions kept, 2 env steps + goal test + is_colliding_ant each] -> accept into the 29-d tree; "
                                      f"{N0}-node snapshot, boxes.csv x 4, seeded random weights; env step: {stand_in}",
                          "batch_per_gpu": Bper, "global_batch": Btot, "calls_per_candidate": nC, "ant_dynamics": dyn,
                          "parallelism": f"candidates sharded x{world}, one {eng.tree.record_doubles * 8}-byte record all-gather per round"},
               "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                            **recorded_traffic("ant_round", prec),
                            "kernel": "all MFMA kernels of the denoiser (3-tap convs at L = 16 / 8 / 4: conv3_halo16x3_kernel, strided / 1x1 / FiLM layers: gemm16_kernel, GroupNorm fused; split formats: 3 MFMAs per product)",
                            "per_kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()},
                            "algorithmic_gflop_per_candidate": 2.0 * mac * nC / 1e9,
                            "note": "events around every MFMA launch inside the timed region (costs a few %)"},
               "outcome": {"ok": int((st == 0).sum()), "goal": int((st == 1).sum()), "collided": int((st == 2).sum()),
                           "mean_chunks_until_end": float(run_chunks.mean())},
               "early_exit": ee, "cpu_baseline": cpu, **comm}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def dataclass_replace(args, **kw):
    import copy
    a = copy.copy(args)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="expand", choices=["expand", "rollout", "mppi", "lidar-round", "ant-round", "geometry"])
    ap.add_argument("--model", default="car", choices=["car", "ant"],
                    help="rollout / mppi workloads: the dynamics behind the rollout-kernel interface.  ant = the build's stand-in "
                         "29-state / 8-action crawler model (NOT MuJoCo, parity unpinned)")
    ap.add_argument("--ant-dynamics", default="model", choices=["model", "tape"],
                    help="ant-round workload: the env step (MuJoCo is not built): the stand-in model or a next-observation tape")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed GLOBAL round size split over the GPUs (strong scaling); default: --batch per GPU (weak)")
    ap.add_argument("--horizon", type=int, default=0, help="rollout workload: steps per rollout (default 16)")
    ap.add_argument("--mppi-lanes", type=int, default=0, choices=[0, 1, 2, 4], help="mppi workload: lanes per rollout (0 = library default)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="candidates per GPU per round")
    ap.add_argument("--precision", default="f16x3", choices=["bf16", "f32", "f16x3", "bf16x3", "f16"],
                    help="denoiser instantiation (include/ditree.h DITREE_PREC_*).  Default f16x3: the fastest one that meets the "
                         "north-star tolerance (flags exact, states 1e-5; tests/test_gpu_round_precision.py)")
    ap.add_argument("--no-throughput-line", action="store_true",
                    help="skip the extra timing of the plain-bf16 instantiation (the throughput mode, off-tolerance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch event timing")
    ap.add_argument("--no-early-exit-line", action="store_true",
                    help="skip the extra (informational) timing with alive-candidate compaction")
    args = ap.parse_args()
    args.batch_set = any(a == "--batch" or a.startswith("--batch=") for a in sys.argv[1:])
    args.precision_set = any(a == "--precision" or a.startswith("--precision=") for a in sys.argv[1:])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # single-command form (`python bench.py --gpus N ...`): start the N ranks from here, before anything touches the GPU
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))
    if os.environ.get("DITREE_BENCH_DRYRUN", "0") == "1":
        return run_dry(args)
    if args.workload == "rollout":
        return run_rollout_ant(args) if args.model == "ant" else run_rollout(args)
    if args.workload == "mppi":
        return run_mppi(args)
    if args.workload == "lidar-round":
        return run_lidar_round(args)
    if args.workload == "ant-round":
        return run_ant_round(args)
    if args.workload == "geometry":
        return run_geometry(args)

    rank, world, local, dist, rehearse = _dist_setup(args)
    force_dist = world == 1 and dist is not None

    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.engine import CNT_GOAL, CNT_LATCH, CNT_NODES, ExpansionEngine
    from ditreeonlineplanner_amd.model import NoisePredNet
    from ditreeonlineplanner_amd.ops import Context

    Bper = args.batch
    Btot = Bper * world
    if args.global_batch:                      # strong scaling: a fixed global round split over the ranks
        if args.global_batch % world:
            raise SystemExit("--global-batch must divide by the number of GPUs")
        Btot = args.global_batch
        Bper = Btot // world
    maze = load_maze("boxes")
    nodes, goal, samples, cond, noise = synth_inputs(maze, Btot)
    ctx = Context(local)
    net = NoisePredNet(seed=0)
    net.bind(ctx, precision=_lib.PREC_NAMES[args.precision], max_batch=Bper)
    eng = ExpansionEngine(ctx, maze, nodes[0], goal, edge_length=H, action_horizon=A, pred_horizon=P, batch=Btot,
                          capacity=N0 + Btot, rank=rank, world_size=world, emulate_sticky_done=False)
    eng.force_allgather = force_dist
    dev = ctx.device
    comm = comm_info(dist, rehearse, dev)
    t = eng.tree
    nd = torch.as_tensor(nodes, device=dev)

    def reset_tree():
        t.counters[CNT_NODES] = N0
        t.counters[CNT_GOAL] = -1
        t.counters[CNT_LATCH] = 0
        t.n_nodes_host = N0
    t.state[:N0] = nd
    t.xy[:N0] = nd[:, :2]
    t.parent[:N0] = torch.arange(-1, N0 - 1, device=dev, dtype=torch.int32).clamp(min=0)
    t.parent[0] = -1
    t.has_prev[1:N0] = 1          # the root has no previous action (as the oracle tree)
    reset_tree()
    s_dev = torch.as_tensor(samples, device=dev)
    c_dev = torch.as_tensor(cond, device=dev)
    n_dev = noise.to(dev)
    torch.cuda.synchronize()

    def step():
        eng.expand_round(s_dev, c_dev, noise=n_dev)       # includes all-gather (N > 1) and accept
        reset_tree()

    for _ in range(args.warmup):
        step()
    if not args.no_profile:
        ctx.profile(2)                      # timed region: events around runs of the dominant kernel only
    if world > 1 or force_dist:
        eng.exchange_events = []            # one event pair per round around pack + all-gather + unpack
    elapsed = _timed(step, dataclass_replace(args, warmup=0), dist, world, dev, rehearse)
    exch_ms = None
    if eng.exchange_events:
        exch_ms = float(np.mean([a.elapsed_time(b) for a, b in eng.exchange_events]))
    eng.exchange_events = None
    prof = prof_all = None
    if not args.no_profile:
        prof = ctx.profile_read(args.precision)
        # second, untimed pass of the same rounds with events around every MFMA launch (bracketing all ~120 launches
        # of a denoiser call costs ~6 %, so it stays out of the timed region): per-kernel times of all three kernels
        ctx.profile(1)
        for _ in range(min(args.steps, 5)):
            step()
        torch.cuda.synchronize()
        prof_all = ctx.profile_read(args.precision)
        prof_all_steps = min(args.steps, 5)
        ctx.profile(0)

    # informational: the same rounds with alive-candidate compaction (collided / finished candidates skip
    # their remaining denoiser calls, as the reference abandons a collided edge).  NOT the headline: `value`
    # above makes every candidate run all H/A denoiser calls.
    ee = None
    if not args.no_early_exit_line:
        eng.early_exit = 1
        e2 = _timed(step, args, dist, world, dev, rehearse)
        run = eng.rb.chunks_run[:Btot].float().mean().item()
        rs = ctx.round_stats()
        q = rs["candidates_per_wave"]
        ee = {"value": Btot * args.steps / e2, "ms_per_step": 1e3 * e2 / args.steps,
              "mean_denoiser_calls_per_candidate": run,
              "speedup_over_value": (Btot * args.steps / e2) / (Btot * args.steps / elapsed),
              "tile_waves_per_layer": {"this_round": rs["tile_waves"], "denoiser_calls": rs["denoiser_calls"],
                                       "candidates_per_wave": q, "without_early_exit": (H // A) * -(-Bper // q),
                                       "one_ragged_call_per_chunk_would_cost": "sum over chunks of ceil(alive / candidates_per_wave)",
                                       "ideal": run * Bper / q},
              "note": "chunks run for alive candidates only, denoiser calls packed to whole tile-waves from the pool of ready "
                      "(candidate, chunk) items (same tree, bit for bit); informational, not comparable with `value`"}
        eng.early_exit = 0

    # The plain bf16 instantiation beside the headline: 3x the rate, but 8 significand bits -- its round-level deviation
    # from the fp32 oracle (profiles/r02_round_precision.json) is outside the north-star tolerance, so it is not `value`.
    tp = None
    if not args.no_throughput_line and args.precision != "bf16":
        net.bind(ctx, precision=_lib.PREC_BF16, max_batch=Bper)
        e3 = _timed(step, dataclass_replace(args, warmup=max(1, args.warmup)), dist, world, dev, rehearse)
        tp = {"dtype": "bf16", "value": Btot * args.steps / e3, "ms_per_step": 1e3 * e3 / args.steps,
              "note": "plain bf16 MFMA inputs: throughput mode, NOT within the north-star tolerance"}
        try:
            # NOT measured in this run: the round-level deviation tests/test_gpu_round_precision.py recorded on MI355X
            with open(os.path.join(REPO, "profiles", "r03_round_precision.json")) as f:
                dv = json.load(f)
            keys = ("max_abs_trajectory_state", "p99_abs_trajectory_state", "flips", "n_agree", "candidates")
            tp["round_deviation_vs_fp32_oracle"] = {"kind": "recorded", "source": "profiles/r03_round_precision.json (config2 round)",
                                                    **{k: dv["bf16"]["config2"][0][k] for k in keys}}
            tp["headline_round_deviation"] = {"kind": "recorded", "source": "profiles/r03_round_precision.json (config2 round)",
                                              **{k: dv[args.precision]["config2"][0][k] for k in keys}}
        except (OSError, KeyError, ValueError, IndexError):
            pass

    if rank == 0:
        n_chunks = H // A
        value = Btot * args.steps / elapsed
        out = {
            "metric": "candidate tree-expansions/sec (carmaze, H=32)", "value": value,
            "unit": "candidate expansions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": args.precision, "data": _data(rehearse),
            "config": {"workload": f"cfgs/carmaze.yaml + fm_policy flow sampler (K=1), batch={Bper} candidates per GPU, "
                                   f"H=32 (4 chunks x 8 steps), boxes.csv 20x20, {N0}-node tree snapshot, seeded random weights",
                       "batch_per_gpu": Bper, "global_batch": Btot, "edge_length": H, "action_horizon": A,
                       "pred_horizon": P, "flow_steps": 1, "tree_nodes": N0, "parallelism": f"candidates sharded x{world}"},
            **comm,
        }
        if exch_ms is not None:
            out["exchange"] = {"ms_per_round": exch_ms, "bytes_per_rank": Bper * 96,
                               "what": "ditree_round_pack + all-gather of 96-byte candidate records + ditree_round_unpack, "
                                       "events on the launch stream inside the timed region (rank 0)"}
        if prof:
            # dominant kernel: timed inside the timed region (the kind the library brackets in mode 2)
            name = max(prof, key=lambda k: prof[k]["ms"])
            d = prof[name]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12            # executed == algorithmic for this kernel (no padding)
            # governing MFMA roofline of the instantiation (MI355X_MICROARCH.md): 16-bit dense 2.5 PFLOP/s; the split
            # instantiations issue 3 MFMAs per algorithmic product, so their ceiling in ALGORITHMIC FLOP/s is a third of it
            peak = {"f32": 157.3, "f16x3": PEAK_BF16_TFLOPS / 3, "bf16x3": PEAK_BF16_TFLOPS / 3}.get(args.precision, PEAK_BF16_TFLOPS)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, **pmc_traffic(name, args.precision), "kernel": name,
                               "launches": d["launches"], "avg_launch_ms": d["ms"] / max(1, d["launches"]),
                               "algorithmic_gflop_per_launch": d["flops"] / max(1, d["launches"]) / 1e9,
                               "kernel_time_share": d["ms"] * 1e-3 / elapsed}
            if prof_all:
                all_ms = sum(v["ms"] for v in prof_all.values())
                all_launches = sum(v["launches"] for v in prof_all.values())
                alg_total = 2.0 * MAC_PER_CALL * Bper * n_chunks * prof_all_steps   # per rank, whole denoiser
                out["roofline"]["all_mfma_kernels"] = {
                    "achieved": alg_total / (all_ms * 1e-3) / 1e12,
                    "frac": alg_total / (all_ms * 1e-3) / 1e12 / peak, "launches": all_launches,
                    "steps": prof_all_steps,
                    "note": "separate untimed pass of the same rounds, events around every MFMA launch: SURVEY 8(d) "
                            "algorithmic FLOPs of the whole denoiser over the summed time of all three MFMA kernels"}
                out["roofline"]["per_kernel_ms_per_step"] = {k: v["ms"] / prof_all_steps for k, v in prof_all.items()}
        if ee is not None:
            out["early_exit"] = ee
        if tp is not None:
            out["throughput_mode"] = tp
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(maze, nodes, goal, samples, cond, noise, net.state_dict())
            # also the faithful sequential loop (B = 1, the reference's own order) on a smaller sample
            seq = cpu_baseline(maze, nodes, goal, samples, cond, noise, net.state_dict(), n_cand=16, batch=1)
            cb["sequential_b1"] = {"value": seq["value"], "unit": seq["unit"], "cores": seq["cores"], "sample": seq["sample"]}
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
