"""Run one of the reference's driver scripts, unchanged, on the MI355X engine:

    python -m ditreeonlineplanner_amd.run /path/to/DiTreeOnlinePlanner/run_scenarios.py [script args...]

`python script.py` puts the script's directory FIRST on sys.path, so the reference's own `planners/`, `policies/`,
`car_env.py` would win over any PYTHONPATH entry.  This launcher puts `dropin/` (engine-backed modules under the
reference's import paths) in front, the script's directory behind it (for everything the engine does not replace:
`planners.MPC`, `obstacle_insertion`, `plot_logger`, `cfgs/`, `maps/`, `metadata/`), changes into the script's
directory (the reference opens `metadata/{env_id}.pt`, `cfgs/*.yaml` relative to the CWD) and executes the script as
`__main__`.

Several GPUs of one node: start one process per GPU,

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ditreeonlineplanner_amd.run script.py

The launcher then binds the process to GPU LOCAL_RANK and joins the RCCL process group before the script starts, and
every `RRT_Planner` the script builds shards its rounds over the ranks (the launcher sets DITREE_SHARD_DEFAULT_GROUP=1:
planners/RRT.py `rank` / `world_size` then default to the group; without it an initialised group is NOT adopted).  The scripts seed every RNG identically on all ranks (run_scenarios.py:86-90), so all ranks draw
the same samples and return the same path."""
import os
import runpy
import sys


def join_process_group():
    """One process per GPU under `torch.distributed.run`: bind to GPU LOCAL_RANK, join the group (nccl = RCCL over xGMI;
    DITREE_REHEARSE_ONE_GPU=1: all ranks on GPU 0 over gloo, tests only).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    rehearse = os.environ.get("DITREE_REHEARSE_ONE_GPU", "0") == "1"
    local = 0 if rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearse:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
    else:
        dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=world, device_id=torch.device("cuda", local))
    # the planners of THIS script shard over the group (engine.default_shard): every rank runs the same script with the same seeds
    os.environ["DITREE_SHARD_DEFAULT_GROUP"] = "1"


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    root = os.path.dirname(script)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dropin = os.path.join(repo, "dropin")
    os.environ["DITREE_REFERENCE_ROOT"] = root
    sys.path[:] = [dropin, repo, root] + [p for p in sys.path if p not in ("", dropin, repo, root)]
    os.chdir(root)
    join_process_group()
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
