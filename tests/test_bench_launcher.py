"""`python bench.py --gpus N` must start by itself: the parent spawns N fresh rank processes, relays rank 0's
one JSON line and fails when a rank fails.  CPU: the launcher + rendezvous in dry-run mode (no GPU, no measurement).
GPU (-m gpu): the real workloads with 2 ranks rehearsed on one GPU over gloo (DITREE_REHEARSE_ONE_GPU)."""
import json
import os
import subprocess
import sys

import pytest

from tests.util import REPO

BENCH = os.path.join(REPO, "bench.py")


def _run(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, BENCH, *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr.decode()


@pytest.mark.parametrize("workload", ["expand", "rollout", "lidar-round"])
def test_launcher_spawns_ranks_and_relays_one_line(workload):
    rc, lines, err = _run(["--gpus", "2", "--steps", "1", "--workload", workload], {"DITREE_BENCH_DRYRUN": "1"})
    assert rc == 0, err
    assert len(lines) == 1, lines                      # exactly one line on stdout, and it is JSON
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dry_run"] is True


def test_launcher_fails_when_a_rank_fails():
    rc, lines, err = _run(["--gpus", "2", "--steps", "1"], {"DITREE_BENCH_DRYRUN": "1", "DITREE_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert rc == 7 and "rank 1 exited with 7" in err


def test_launcher_refuses_more_ranks_than_gpus():
    # this container has no GPU: the parent must say so (rc 2) instead of starting ranks that cannot get a device
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    rc, lines, err = _run(["--gpus", "2", "--steps", "1"], {})
    assert rc == 2 and not lines and "GPU(s) visible" in err


@pytest.mark.gpu
def test_bench_two_ranks_rehearsed_on_one_gpu():
    rc, lines, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256", "--no-cpu-baseline",
                           "--no-throughput-line", "--no-early-exit-line"], {"DITREE_REHEARSE_ONE_GPU": "1"}, timeout=900)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["global_batch"] == 512
    assert out["exchange"]["ms_per_round"] > 0 and out["exchange"]["bytes_per_rank"] == 256 * 96
    assert "rehearsal" in out["data"] and out["value"] > 0


@pytest.mark.gpu
def test_bench_lidar_round_two_ranks_rehearsed_on_one_gpu():
    rc, lines, err = _run(["--workload", "lidar-round", "--global-batch", "8192", "--gpus", "2", "--steps", "1", "--warmup", "1"],
                          {"DITREE_REHEARSE_ONE_GPU": "1"}, timeout=900)
    assert rc == 0, err[-3000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["batch_per_gpu"] == 4096
    assert out["scaling"] == "strong" and out["exchange"]["ms_per_round"] > 0


@pytest.mark.gpu
def test_bench_mppi_two_ranks_rehearsed_on_one_gpu():
    """BASELINE config 5 sharded: the K global rollouts split over 2 ranks, two all-reduces per controller step."""
    rc, lines, err = _run(["--workload", "mppi", "--global-batch", "8192", "--gpus", "2", "--steps", "20", "--warmup", "3"],
                          {"DITREE_REHEARSE_ONE_GPU": "1"}, timeout=600)
    assert rc == 0, err[-3000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["rollouts_per_gpu"] == 4096
    assert out["scaling"] == "strong" and out["value"] > 0 and out["roofline"]["fp64"]["achieved_tflops"] > 0


@pytest.mark.gpu
def test_bench_rccl_path_on_one_gpu():
    """DITREE_FORCE_DIST=1: the record exchange through the nccl backend (RCCL) with a one-rank group -- the collective, its
    event timing and `ranks_seen` run on the real backend, which the gloo rehearsals cannot show."""
    rc, lines, err = _run(["--steps", "2", "--warmup", "1", "--batch", "256", "--no-cpu-baseline", "--no-throughput-line",
                           "--no-early-exit-line"], {"DITREE_FORCE_DIST": "1"}, timeout=600)
    assert rc == 0, err[-3000:]
    out = json.loads(lines[-1])
    assert out["ranks_seen"] == 1 and out["backend"].startswith("nccl") and out["exchange"]["ms_per_round"] > 0
    # the driver's contract for the line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["metric"].startswith("candidate tree-expansions/sec (carmaze, H=32)") and out["vs_baseline"] is None
    assert out["dtype"] == "f16x3" and "workload" in out["config"] and out["higher_is_better"] is True
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in r, k
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel"].startswith("conv3_halo16x3_kernel")


def test_launcher_takes_its_ranks_down_when_it_is_terminated(tmp_path):
    """SIGTERM to the launcher (a driver's timeout) must end the rank processes it started, not orphan them."""
    import signal
    import time
    env = dict(os.environ, DITREE_BENCH_DRYRUN="1", DITREE_BENCH_DRYRUN_SLEEP="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    kids = []
    for _ in range(100):                                   # wait until both ranks exist
        out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], stdout=subprocess.PIPE).stdout.split()
        kids = [int(x) for x in out]
        if len(kids) >= 2:
            break
        time.sleep(0.1)
    assert len(kids) >= 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for k in kids:
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z", k


@pytest.mark.gpu
def test_bench_ant_workloads_print_labelled_lines():
    """`--workload ant-round` (BASELINE config 3 as a real round) with 2 ranks rehearsed on one GPU, and `--workload rollout
    --model ant` (the higher-DoF rollout slot): one JSON line each, the stand-in dynamics named in metric / data / config."""
    rc, lines, err = _run(["--workload", "ant-round", "--gpus", "2", "--batch", "128", "--steps", "1", "--warmup", "0",
                           "--no-cpu-baseline", "--no-early-exit-line"], {"DITREE_REHEARSE_ONE_GPU": "1"}, timeout=900)
    assert rc == 0, err[-3000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["global_batch"] == 256 and out["value"] > 0
    assert "NOT MuJoCo" in out["metric"] and "NOT MuJoCo" in out["data"] and out["config"]["ant_dynamics"] == "model"
    assert out["roofline"]["bound"] == "mfma" and out["outcome"]["ok"] + out["outcome"]["goal"] + out["outcome"]["collided"] == 128
    assert out["exchange"]["bytes_per_rank"] == 128 * 134 * 8
    rc, lines, err = _run(["--workload", "rollout", "--model", "ant", "--batch", "8192", "--steps", "3", "--warmup", "1"], {}, timeout=600)
    assert rc == 0, err[-3000:]
    out = json.loads(lines[-1])
    assert "NOT MuJoCo" in out["metric"] and out["roofline"]["bound"] == "hbm" and out["roofline"]["kernel"] == "ant_rollout_kernel<true>"
    assert out["roofline"]["algorithmic_bytes_per_launch"] == 8192 * (232 + 64 * 16 + 232 * 16 + 8) and out["value"] > 0
