"""CPU: the oracle's ResNet-18-GN encoder against an INDEPENDENT implementation of the same published architecture.

The reference builds its map encoder from `torchvision.models.resnet18()` + `replace_bn_with_gn` (local_map_encoder.py:63-76,
111-122).  torchvision is absent here, so the oracle restates the topology (oracle/denoiser.py `_ResNet18GN`) and SURVEY.md
8(c) / DESIGN.md call that piece "parity unpinned": nothing the reference holds can check it.  This test does not change
that status -- it is not the reference -- but it removes the risk of a private misreading of the architecture: Hugging Face
`transformers` ships its own ResNet (`ResNetModel`, basic layers, depths 2-2-2-2 = ResNet-18; written independently of
torchvision), which is importable in this image.  With the oracle's weights copied over and every BatchNorm2d replaced by
GroupNorm(C // 16, C) -- the reference's `replace_bn_with_gn` rule -- both must give the same embedding."""
import numpy as np
import pytest
import torch
from torch import nn

from oracle import denoiser as OD


def _hf_resnet18_gn():
    tr = pytest.importorskip("transformers")
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                          layer_type="basic", hidden_act="relu", downsample_in_first_stage=False)
    m = tr.ResNetModel(cfg).eval()

    def swap(mod):                       # local_map_encoder.py:63-76: GroupNorm(num_features // 16, num_features)
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.BatchNorm2d):
                setattr(mod, name, nn.GroupNorm(child.num_features // 16, child.num_features))
            else:
                swap(child)
    swap(m)
    assert not any(isinstance(x, nn.BatchNorm2d) for x in m.modules())
    return m


def _copy_weights(oracle_resnet, hf):
    """torchvision names (the oracle's state-dict keys) -> transformers names."""
    src = oracle_resnet.state_dict()
    dst = hf.state_dict()
    put = {}

    def norm_conv(o_conv, o_norm, h_prefix):
        put[h_prefix + ".convolution.weight"] = src[o_conv + ".weight"]
        put[h_prefix + ".normalization.weight"] = src[o_norm + ".weight"]
        put[h_prefix + ".normalization.bias"] = src[o_norm + ".bias"]
    norm_conv("conv1", "bn1", "embedder.embedder")
    for li in range(4):
        for bi in range(2):
            o = f"layer{li + 1}.{bi}"
            h = f"encoder.stages.{li}.layers.{bi}"
            norm_conv(o + ".conv1", o + ".bn1", h + ".layer.0")
            norm_conv(o + ".conv2", o + ".bn2", h + ".layer.1")
            if o + ".downsample.0.weight" in src:
                norm_conv(o + ".downsample.0", o + ".downsample.1", h + ".shortcut")
    assert set(put) == set(dst), (set(dst) - set(put), set(put) - set(dst))
    for k, v in put.items():
        assert dst[k].shape == v.shape, (k, dst[k].shape, v.shape)
    hf.load_state_dict(put)


@pytest.mark.parametrize("n", [20, 16])          # the car's and the ant's local map
def test_oracle_encoder_equals_an_independent_resnet18_gn(n):
    torch.manual_seed(0)
    net = OD.init_noise_pred_net().eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                        # non-trivial GroupNorm affines
        for name, p in net.named_parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn(p.shape, generator=g))
    enc = net.encoder
    hf = _hf_resnet18_gn()
    _copy_weights(enc.resnet18, hf)
    lm = (torch.rand(12, n, n, generator=g) < 0.3).float() * 2 - 1
    with torch.no_grad():
        ref = enc(lm)                                                     # (B, 400)
        x = lm.unsqueeze(1).repeat(1, 3, 1, 1)                             # local_map_encoder.py:117-119
        pooled = hf(pixel_values=x).pooler_output.flatten(1)              # conv stem, 4 stages, adaptive average pool
        got = enc.resnet18.fc(pooled)
    assert ref.shape == got.shape == (12, 400)
    assert np.abs(ref.numpy() - got.numpy()).max() < 1e-4 * max(1.0, float(ref.abs().max()))
    assert float((ref - got).norm() / ref.norm()) < 1e-5
