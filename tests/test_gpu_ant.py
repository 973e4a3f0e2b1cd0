"""BASELINE config 3 (cfgs/antmaze.yaml + fm_policy): the ant-sized denoiser and its glue on the GPU.

The ant's DYNAMICS (MuJoCo through gymnasium-robotics) have no oracle here and are not built; everything
around them is (the round itself: tests/test_gpu_ant_round.py): sampler pre-processing incl. quaternion -> rot6d (policies/fm_policy.py:77-82, common/se3_utils.py:177-189),
the 16 x 16 @ 0.8 local map with s_global = 4 (run_scenarios.py:123-132; tests/test_gpu_geometry.py), and the denoiser at
input_dim 8, pred_horizon 16, cond 97 (+ 400).  The oracle's U-Net at these dimensions is bit-identical to the reference class
(golden `unet_ant_*`, tests/test_oracle_golden.py); its sampler pre-processing equals the reference on the `sampler_ant_*` cases."""
import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import sampler as OS
from tests.util import golden

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    return t if dtype is None else t.to(dtype)


@pytest.fixture(scope="module")
def ctx():
    from ditreeonlineplanner_amd.ops import Context
    c = Context(0)
    yield c
    c.close()


def ant_norm():
    m = OS.ANT_META
    return np.concatenate([m["Observations_mean"], m["Observations_std"], m["Actions_mean"], m["Actions_std"]])


def test_cond_vector_ant_matches_reference_cases(ctx):
    g = golden("network")
    obs, nh = g["sampler_ant_obs"], g["sampler_ant_n_hist"]
    for h in (1, 2, 3):                     # the kernel takes one history length per launch
        rows = np.nonzero(nh == h)[0]
        if rows.size == 0:
            continue
        out = ctx.cond_vector_ant(dev(obs[rows][:, 3 - h:]), dev(g["sampler_ant_prev"][rows]),
                                  dev(g["sampler_ant_has_prev"][rows].astype(np.uint8)), dev(g["sampler_ant_goals"][rows]),
                                  16, ant_norm()).cpu().numpy()
        exp = g["sampler_ant_cond_expected"][rows]
        assert np.abs(out - exp).max() < 2e-6, (h, np.abs(out - exp).max())
        if h < 3:                           # missing history steps are zero rows in front
            assert np.array_equal(out[:, : (3 - h) * 29], np.zeros((rows.size, (3 - h) * 29), dtype=np.float32))
        nop = ~g["sampler_ant_has_prev"][rows]
        assert np.array_equal(out[nop, 87:95], np.zeros((int(nop.sum()), 8), dtype=np.float32))
    with pytest.raises(ValueError):
        ctx.cond_vector_ant(dev(obs[:2, :, :28].copy()), dev(g["sampler_ant_prev"][:2]), dev(np.ones(2, dtype=np.uint8)),
                            dev(g["sampler_ant_goals"][:2]), 16, ant_norm())


@pytest.fixture(scope="module")
def ant_net():
    torch.manual_seed(0)
    net = OD.init_noise_pred_net(input_dim=8, action_dim=8, obs_dim=29, obs_history=3, action_history=1).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    return net


# Per-layer bounds as for the car network (tests/test_gpu_denoiser.py): relative L2, max|err| / rms and an element-wise bound
# with no violation allowed, per tapped layer and on the flow-step output.  The ant levels L = 8 and L = 4 run other kernels
# than the car's (the SHORT epilogue of gemm16_kernel / conv3_halo16x3_kernel on the 16-bit tiles; conv_gemm_kernel +
# gn1d_short_kernel for f32 and for batches that do not fill a tile), so they carry their own measured numbers:
# gpurun_out/ant_layers_prec*.json (committed as profiles/r03_ant_layers.json); bounds = the car's, times ANT_SCALE.
from tests.test_gpu_denoiser import LAYERS, TOL, _oracle_with_taps, check_close, err_stats, tap_to_blc  # noqa: E402

ANT_SCALE = {1: 1.0, 2: 1.0, 3: 1.0, 4: 1.0, 0: 1.0}      # measured (profiles/r03_ant_layers.json): at or below the car network's errors


def ant_tol(prec):
    return {k: v * ANT_SCALE[prec] for k, v in TOL[prec].items()}


def _ant_inputs():
    g = torch.Generator().manual_seed(21)
    B = 24
    noise = torch.randn(B, 16, 8, generator=g)
    cond = torch.randn(B, 97, generator=g) * 0.6
    lm = (torch.rand(B, 16, 16, generator=g) < 0.3).float() * 2 - 1
    return B, noise, lm, cond


def _ant_bind(ctx, ant_net, prec, B):
    from ditreeonlineplanner_amd.model import NoisePredNet
    net = NoisePredNet(input_dim=8, additional_global_cond_dim=97, pred_horizon=16, local_map_size=16, init=False)
    net.load_state_dict(ant_net.state_dict())
    net.bind(ctx, precision=prec, max_batch=B)
    return net


@pytest.mark.parametrize("prec", [1, 2, 3, 4, 0])
def test_ant_denoiser_layers_and_output(ctx, ant_net, prec):
    import json
    import os
    from tests.util import REPO
    B, noise, lm, cond = _ant_inputs()
    _ant_bind(ctx, ant_net, prec, B)
    assert ctx.denoise_dims() == (16, 8, 16, 97, 400)
    x_ref, taps = _oracle_with_taps(ant_net, noise, lm, cond)
    unit = np.concatenate([np.zeros(8), np.ones(8)])
    x = ctx.denoise(noise.cuda(), lm.cuda(), cond.cuda(), act_norm=unit, want_actions=False).cpu().numpy()
    tol = ant_tol(prec)
    report, bad = {}, []
    for name, _ in LAYERS:
        if name == "enc.pool":
            continue
        got = ctx.debug_read(name, B).cpu().numpy()
        ref = tap_to_blc(taps[name].numpy(), B)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        report[name] = err_stats(got, ref)
        bad += check_close(got, ref, tol, name)
    report["x1"] = err_stats(x, x_ref)
    bad += check_close(x, x_ref, tol, "x1")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", f"ant_layers_prec{prec}.json"), "w") as f:
        json.dump(report, f, indent=1)
    assert not bad, bad
    # one corrupted element per layer must be seen (the L = 8 / 4 levels included)
    rng = np.random.default_rng(7)
    for name in ("d0b1.out", "skip1", "d2b1.out", "mid2.out", "u0b2.out", "final.in"):
        got = ctx.debug_read(name, B).cpu().numpy()
        ref = tap_to_blc(taps[name].numpy(), B)
        rms = float(np.sqrt(np.mean(ref ** 2)))
        flat = got.reshape(-1).copy()
        for _ in range(1000):
            i, j = rng.integers(0, flat.size, 2)
            if abs(flat[i] - flat[j]) >= rms:
                break
        flat[i] = flat[j]
        assert check_close(flat.reshape(got.shape), ref, tol, name), (name, "corruption not detected")


@pytest.mark.parametrize("prec", [2, 0])
def test_ant_actions_use_every_dimension_of_the_statistics(ctx, ant_net, prec):
    """The D = 8 un-normalisation a = x * std + mean (policies/fm_policy.py:201-203) with a NON-trivial, per-dimension mean
    and std (metadata/antmaze.pt has mean 0 / std 1, which a swapped or mis-strided table would survive), a ragged
    sub-batch, and shape checks."""
    B, noise, lm, cond = _ant_inputs()
    _ant_bind(ctx, ant_net, prec, B)
    rng = np.random.default_rng(5)
    mean, std = rng.normal(0.0, 1.0, 8), rng.uniform(0.5, 2.0, 8)
    unit = np.concatenate([np.zeros(8), np.ones(8)])
    x = ctx.denoise(noise.cuda(), lm.cuda(), cond.cuda(), act_norm=unit, want_actions=False).cpu().numpy()
    a = ctx.denoise(noise[:5].cuda().contiguous(), lm[:5].cuda().contiguous(), cond[:5].cuda().contiguous(),
                    act_norm=np.concatenate([mean, std]), want_actions=True)
    assert a.dtype == torch.float64 and a.shape == (5, 16, 8)
    exp = x[:5].astype(np.float64) * std + mean                    # float32 * float64 -> float64, as the reference
    assert np.abs(a.cpu().numpy() - exp).max() < 1e-12
    with pytest.raises(ValueError):                                 # mean / std of the wrong width
        ctx.denoise(noise.cuda(), lm.cuda(), cond.cuda(), act_norm=np.ones(4), want_actions=True)
    with pytest.raises(ValueError):                                 # wrong shapes are refused, not mis-strided
        ctx.denoise(torch.zeros(B, 64, 2, device="cuda"), lm.cuda(), cond.cuda(), want_actions=False)


def test_ant_sampler_facade(ctx, ant_net):
    """DiffusionSampler(env_id='antmaze') end to end = explicit pre-processing + denoiser with the same device RNG state."""
    from ditreeonlineplanner_amd._lib import PREC_F32
    from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler
    from ditreeonlineplanner_amd.train_diffusion_policy import init_noise_pred_net
    net = init_noise_pred_net(input_dim=8, action_dim=8, obs_dim=29, obs_history=3, action_history=1, goal_conditioned=True,
                              goal_dim=2, local_map_conditioned=True, local_map_encoder="resnet", local_map_embedding_dim=400,
                              local_map_size=16, down_dims=[512, 1024, 2048])
    net.load_state_dict(ant_net.state_dict())
    smp = DiffusionSampler(net, None, "antmaze", policy="flow_matching", pred_horizon=16, action_dim=8, prediction_type="actions",
                           obs_history=3, action_history=1, goal_conditioned=True, num_diffusion_iters=1, local_map_size=16,
                           ctx=ctx, precision=PREC_F32)
    gq = golden("network")
    obs = gq["sampler_ant_obs"][:6]
    prev = gq["sampler_ant_prev"][:6]
    goal = gq["sampler_ant_goals"][0]
    lm01 = (np.random.default_rng(3).random((6, 16, 16)) < 0.3).astype(np.float32)
    torch.manual_seed(77)
    a = smp(obs, prev_actions=prev[:, None, :], goal=goal, local_map=lm01)
    assert a.shape == (6, 16, 8) and a.dtype == np.float64 and np.isfinite(a).all()
    torch.manual_seed(77)
    noise = torch.randn((6, 16, 8), device="cuda")
    cond = OS.ant_cond_vector(obs, prev, np.ones(6, dtype=bool), goal)
    x_ref = OS.flow_sample(ant_net, noise.cpu(), lm01 * 2 - 1, cond, k_steps=1)
    a_ref = x_ref.astype(np.float64) * OS.ANT_META["Actions_std"] + OS.ANT_META["Actions_mean"]
    assert np.abs(a - a_ref).max() < 1e-4, np.abs(a - a_ref).max()


# The ant ROUND (tree, collision / goal tests, accept) is tests/test_gpu_ant_round.py.
