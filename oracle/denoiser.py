"""torch-CPU fp32 restatement of the reference denoiser (encoder + FiLM 1-D U-Net).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference sites:
  model/diffusion/conditional_unet1d.py:41-142 (ConditionalResidualBlock1D, FiLM),
      :145-266 (constructor, module creation ORDER = parameter-init RNG order),
      :268-347 (forward: time MLP, down / mid / up with skips, final conv)
  model/diffusion/conv1d_components.py:7-40 (Downsample1d, Upsample1d, Conv1dBlock)
  model/diffusion/positional_embedding.py:5-17 (SinusoidalPosEmb)
  local_map_encoder.py:63-76 (BatchNorm -> GroupNorm(C/16)), :78-109 (wrapper),
      :112-122 (ResNet18Encoder: repeat to 3 channels, resnet18, fc 512 -> E)
  train_diffusion_policy.py:32-67 (init_noise_pred_net: global_cond_dim arithmetic)

Parameter names equal the reference's state-dict keys (``unet.*``,
``encoder.resnet18.*``) so a reference checkpoint's ``noise_pred_net_state_dict``
loads with ``load_state_dict`` unchanged.  The U-Net half creates its sub-modules in
the reference's constructor order, so ``torch.manual_seed(s)`` followed by
construction yields bit-identical weights to the reference class (verified by
tests/golden/make_golden.py when the goldens are generated).  torchvision is absent
from the build container, so the ResNet-18 half follows the published torchvision
0.22 topology and is **parity unpinned** against the reference.  It is cross-checked
against an independent implementation of the same architecture (transformers'
ResNetModel with the BatchNorm -> GroupNorm(C/16) rule applied):
tests/test_oracle_encoder_crosscheck.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------ U-Net
class _Seq(nn.Module):
    """Holds children under integer names like nn.Sequential, without a forward."""

    def __init__(self, **mods):
        super().__init__()
        for k, m in mods.items():
            self.add_module(k, m)


def _conv_block(cin, cout, k, groups=8):
    # Conv1dBlock: self.block = Sequential(Conv1d, GroupNorm, Mish) -> keys block.0 / block.1
    blk = nn.Module()
    blk.block = _Seq(**{"0": nn.Conv1d(cin, cout, k, padding=k // 2), "1": nn.GroupNorm(groups, cout)})
    return blk


def _run_conv_block(blk, x):
    conv, gn = blk.block._modules["0"], blk.block._modules["1"]
    return F.mish(gn(conv(x)))


class _CRB(nn.Module):
    """ConditionalResidualBlock1D with condition_type='film' (conditional_unet1d.py:41-142)."""

    def __init__(self, cin, cout, cond_dim, k=3, groups=8):
        super().__init__()
        self.blocks = nn.ModuleList([_conv_block(cin, cout, k, groups), _conv_block(cout, cout, k, groups)])
        self.cond_encoder = _Seq(**{"1": nn.Linear(cond_dim, cout * 2)})      # Sequential(Mish, Linear, Rearrange)
        self.out_channels = cout
        self.residual_conv = nn.Conv1d(cin, cout, 1) if cin != cout else nn.Identity()

    def forward(self, x, cond):
        out = _run_conv_block(self.blocks[0], x)
        emb = self.cond_encoder._modules["1"](F.mish(cond))
        emb = emb.reshape(emb.shape[0], 2, self.out_channels, 1)
        out = emb[:, 0] * out + emb[:, 1]
        out = _run_conv_block(self.blocks[1], out)
        return out + self.residual_conv(x)


class _Down(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class _Up(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)


def sinusoidal_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """positional_embedding.py:10-17."""
    half = dim // 2
    w = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -w)
    e = t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


class OracleUnet1D(nn.Module):
    """ConditionalUnet1D(input_dim, global_cond_dim, 256, down_dims, 3, 8, 'film')."""

    def __init__(self, input_dim, global_cond_dim, dsed=256, down_dims=(512, 1024, 2048), k=3, groups=8):
        super().__init__()
        dims = [input_dim] + list(down_dims)
        # creation order follows conditional_unet1d.py:176-260
        self.dsed = dsed
        step_enc = _Seq(**{"1": nn.Linear(dsed, dsed * 4), "3": nn.Linear(dsed * 4, dsed)})
        cond_dim = dsed + global_cond_dim
        in_out = list(zip(dims[:-1], dims[1:]))
        mid = dims[-1]
        self.mid_modules = nn.ModuleList([_CRB(mid, mid, cond_dim, k, groups), _CRB(mid, mid, cond_dim, k, groups)])
        down = nn.ModuleList()
        for ind, (di, do) in enumerate(in_out):
            last = ind >= len(in_out) - 1
            down.append(nn.ModuleList([_CRB(di, do, cond_dim, k, groups), _CRB(do, do, cond_dim, k, groups),
                                       _Down(do) if not last else nn.Identity()]))
        up = nn.ModuleList()
        for ind, (di, do) in enumerate(reversed(in_out[1:])):
            last = ind >= len(in_out) - 1          # never true: both up stages upsample (:239-251)
            up.append(nn.ModuleList([_CRB(do * 2, di, cond_dim, k, groups), _CRB(di, di, cond_dim, k, groups),
                                     _Up(di) if not last else nn.Identity()]))
        start = down_dims[0]
        final = _Seq(**{"0": _conv_block(start, start, k), "1": nn.Conv1d(start, input_dim, 1)})
        self.diffusion_step_encoder = step_enc
        self.up_modules = up
        self.down_modules = down
        self.final_conv = final

    def time_embedding(self, timestep, batch):
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(batch)
        e = sinusoidal_embedding(t, self.dsed)
        enc = self.diffusion_step_encoder._modules
        return enc["3"](F.mish(enc["1"](e)))

    def forward(self, sample, timestep, global_cond):
        x = sample.permute(0, 2, 1)                                     # b h t -> b t h
        feat = torch.cat([self.time_embedding(timestep, sample.shape[0]), global_cond], dim=-1)
        skips = []
        for r1, r2, ds in self.down_modules:
            x = r2(r1(x, feat), feat)
            skips.append(x)
            x = ds.conv(x) if isinstance(ds, _Down) else x
        for m in self.mid_modules:
            x = m(x, feat)
        for r1, r2, us in self.up_modules:
            x = torch.cat((x, skips.pop()), dim=1)
            x = r2(r1(x, feat), feat)
            x = us.conv(x) if isinstance(us, _Up) else x
        fc = self.final_conv._modules
        x = fc["1"](_run_conv_block(fc["0"], x))
        return x.permute(0, 2, 1)


# ------------------------------------------------------------------------------ encoder
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.GroupNorm(cout // 16, cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.GroupNorm(cout // 16, cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = _Seq(**{"0": nn.Conv2d(cin, cout, 1, stride, bias=False),
                                      "1": nn.GroupNorm(cout // 16, cout)})

    def forward(self, x):
        idt = x
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            d = self.downsample._modules
            idt = d["1"](d["0"](x))
        return F.relu(out + idt)


class _ResNet18GN(nn.Module):
    def __init__(self, embedding_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.GroupNorm(64 // 16, 64)
        chans = [64, 128, 256, 512]
        cin = 64
        for li, c in enumerate(chans):
            stride = 1 if li == 0 else 2
            layer = nn.Sequential(_BasicBlock(cin, c, stride), _BasicBlock(c, c, 1))
            setattr(self, f"layer{li + 1}", layer)
            cin = c
        for m in self.modules():                      # torchvision resnet.py init loop
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.fc = nn.Linear(512, embedding_dim)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(x)


class _Encoder(nn.Module):
    def __init__(self, embedding_dim):
        super().__init__()
        self.resnet18 = _ResNet18GN(embedding_dim)

    def forward(self, local_map):
        x = local_map.unsqueeze(1).repeat(1, 3, 1, 1)          # local_map_encoder.py:119-120
        return self.resnet18(x)


class OracleNoisePredNet(nn.Module):
    """ConditionalUnet1DWithLocalMap(encoder_name='resnet'), local_map_encoder.py:78-109."""

    def __init__(self, input_dim=2, embedding_dim=400, additional_global_cond_dim=7,
                 down_dims=(512, 1024, 2048)):
        super().__init__()
        self.encoder = _Encoder(embedding_dim)
        self.unet = OracleUnet1D(input_dim, embedding_dim + additional_global_cond_dim, down_dims=down_dims)

    def forward(self, sample, local_map, timestep, global_cond=None):
        emb = self.encoder(local_map)
        gc = emb if global_cond is None else torch.cat([emb, global_cond], dim=1)
        return self.unet(sample, timestep, gc)


def init_noise_pred_net(input_dim=2, action_dim=2, obs_dim=3, obs_history=1, action_history=1,
                        goal_conditioned=True, goal_dim=2, local_map_embedding_dim=400,
                        down_dims=(512, 1024, 2048)):
    """train_diffusion_policy.py:32-67 for local_map_encoder='resnet'."""
    gcd = obs_dim * obs_history + goal_dim * int(goal_conditioned) + action_history * action_dim
    return OracleNoisePredNet(input_dim, local_map_embedding_dim, gcd, down_dims)
