"""N > 1 path.  CPU (gloo, world_size 2): the fixed-stride all-gather of candidate records and the
candidate partition.  GPU (-m gpu, 2 ranks sharing cuda:0 over gloo): a sharded round builds the
same tree, bit for bit, as the single-rank round."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import REPO, load_maze


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, B):
    sys.path.insert(0, REPO)
    from ditreeonlineplanner_amd.engine import allgather_round_fields
    _init(rank, world, port)
    per = (B + world - 1) // world
    total = per * world
    ref_i = torch.arange(total, dtype=torch.int32) * 3 + 1
    ref_f = torch.arange(total * 6, dtype=torch.float64).reshape(total, 6) * 0.5
    fi = torch.full((total,), -7, dtype=torch.int32)
    ff = torch.full((total, 6), -7.0, dtype=torch.float64)
    fi[rank * per:(rank + 1) * per] = ref_i[rank * per:(rank + 1) * per]
    ff[rank * per:(rank + 1) * per] = ref_f[rank * per:(rank + 1) * per]
    allgather_round_fields([fi, ff], per, rank, world)
    assert torch.equal(fi, ref_i) and torch.equal(ff, ref_f)
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 13])
def test_allgather_fields_gloo_cpu(B):
    mp.spawn(_cpu_worker, args=(2, 29531 + B, B), nprocs=2, join=True)


def test_shard_partition():
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    for world in (1, 2, 4, 8):
        for B in (1, 7, 1024, 8190):
            got = []
            for r in range(world):
                e = ExpansionEngine.__new__(ExpansionEngine)
                e.rank, e.world = r, world
                lo, hi, per = e.shard(B)
                got.extend(range(lo, hi))
                assert hi - lo <= per
            assert got == list(range(B))


def _gpu_worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    _init(rank, world, port)
    from ditreeonlineplanner_amd.engine import ExpansionEngine
    from ditreeonlineplanner_amd.ops import Context
    from oracle import geometry as G
    from oracle import rrt as ORRT
    from oracle.tapes import ActionTape
    maze = load_maze("boxes")
    start = np.array([*G.cell_rowcol_to_xy([17, 2], maze), np.deg2rad(45.0), 0, 0, 0])
    goal = np.array([*G.cell_rowcol_to_xy([2, 17], maze), 0, 0, 0, 0])
    ctx = Context(0)
    B = 96
    eng = ExpansionEngine(ctx, maze, start, goal, batch=B, capacity=4096, rank=rank, world_size=world)
    rt, at = ORRT.RandomTape(42), ActionTape(7)
    done = 0
    for _ in range(5):
        s, c = rt.draw_round(B, maze.shape[1], maze.shape[0], goal)
        acts = np.stack([at.actions(np.arange(done, done + B), j) for j in range(eng.n_chunks)], axis=1)
        eng.expand_round(torch.as_tensor(s).cuda(), torch.as_tensor(c).cuda(), inject_actions=torch.as_tensor(acts).cuda())
        done += B
    snap = eng.tree_snapshot()
    np.savez(out_path.format(rank=rank), parents=snap["parents"], states=snap["states"], counters=snap["counters"])
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_round_builds_identical_tree(tmp_path):
    out = str(tmp_path / "w{w}_r{rank}.npz")
    mp.spawn(_gpu_worker, args=(1, 29611, out.replace("{w}", "1")), nprocs=1, join=True)
    mp.spawn(_gpu_worker, args=(2, 29612, out.replace("{w}", "2")), nprocs=2, join=True)
    a = np.load(out.replace("{w}", "1").format(rank=0))
    for r in range(2):
        b = np.load(out.replace("{w}", "2").format(rank=r))
        assert np.array_equal(a["parents"], b["parents"])
        assert np.array_equal(a["states"], b["states"])          # same kernels, same inputs: bit-identical
        assert np.array_equal(a["counters"][:5], b["counters"][:5])
    assert len(a["parents"]) > 50
