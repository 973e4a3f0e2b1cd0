"""State-dict <-> flat blob + manifest for ``ditree_load_weights`` (include/ditree.h).

The reference stores weights as ``torch.save({'noise_pred_net_state_dict': ...})``
(train_diffusion_policy.py:468-476) and loads them with ``load_state_dict``
(run_scenarios.py:175-177); key prefixes ``encoder.resnet18.*`` and ``unet.*``.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch


def unet_param_shapes(input_dim=2, global_cond_dim=407, down_dims=(512, 1024, 2048), dsed=256, k=3):
    """Parameter names/shapes of ConditionalUnet1D (model/diffusion/conditional_unet1d.py:145-266)."""
    sh = OrderedDict()
    cond = dsed + global_cond_dim

    def lin(name, i, o):
        sh[f"{name}.weight"] = (o, i)
        sh[f"{name}.bias"] = (o,)

    def conv(name, i, o, kk):
        sh[f"{name}.weight"] = (o, i, kk)
        sh[f"{name}.bias"] = (o,)

    def block(name, i, o):
        conv(f"{name}.block.0", i, o, k)
        sh[f"{name}.block.1.weight"] = (o,)
        sh[f"{name}.block.1.bias"] = (o,)

    def crb(name, i, o):
        block(f"{name}.blocks.0", i, o)
        block(f"{name}.blocks.1", o, o)
        lin(f"{name}.cond_encoder.1", cond, 2 * o)
        if i != o:
            conv(f"{name}.residual_conv", i, o, 1)

    lin("diffusion_step_encoder.1", dsed, dsed * 4)
    lin("diffusion_step_encoder.3", dsed * 4, dsed)
    dims = [input_dim] + list(down_dims)
    in_out = list(zip(dims[:-1], dims[1:]))
    mid = dims[-1]
    crb("mid_modules.0", mid, mid)
    crb("mid_modules.1", mid, mid)
    for ind, (di, do) in enumerate(in_out):
        crb(f"down_modules.{ind}.0", di, do)
        crb(f"down_modules.{ind}.1", do, do)
        if ind < len(in_out) - 1:
            conv(f"down_modules.{ind}.2.conv", do, do, 3)
    for ind, (di, do) in enumerate(reversed(in_out[1:])):
        crb(f"up_modules.{ind}.0", do * 2, di)
        crb(f"up_modules.{ind}.1", di, di)
        sh[f"up_modules.{ind}.2.conv.weight"] = (di, di, 4)            # ConvTranspose1d: (in, out, k)
        sh[f"up_modules.{ind}.2.conv.bias"] = (di,)
    block("final_conv.0", down_dims[0], down_dims[0])
    conv("final_conv.1", down_dims[0], input_dim, 1)
    return sh


def resnet18_gn_param_shapes(embedding_dim=400):
    """torchvision resnet18 with BatchNorm -> GroupNorm and fc -> Linear(512, E)
    (local_map_encoder.py:63-76,112-117)."""
    sh = OrderedDict()
    sh["conv1.weight"] = (64, 3, 7, 7)
    sh["bn1.weight"] = (64,)
    sh["bn1.bias"] = (64,)
    cin = 64
    for li, c in enumerate((64, 128, 256, 512)):
        for bi in range(2):
            p = f"layer{li + 1}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            sh[f"{p}.conv1.weight"] = (c, cin if bi == 0 else c, 3, 3)
            sh[f"{p}.bn1.weight"] = (c,)
            sh[f"{p}.bn1.bias"] = (c,)
            sh[f"{p}.conv2.weight"] = (c, c, 3, 3)
            sh[f"{p}.bn2.weight"] = (c,)
            sh[f"{p}.bn2.bias"] = (c,)
            if bi == 0 and (stride != 1 or cin != c):
                sh[f"{p}.downsample.0.weight"] = (c, cin, 1, 1)
                sh[f"{p}.downsample.1.weight"] = (c,)
                sh[f"{p}.downsample.1.bias"] = (c,)
        cin = c
    sh["fc.weight"] = (embedding_dim, 512)
    sh["fc.bias"] = (embedding_dim,)
    return sh


def noise_pred_net_param_shapes(input_dim=2, embedding_dim=400, additional_global_cond_dim=7,
                                down_dims=(512, 1024, 2048)):
    sh = OrderedDict()
    for k, v in resnet18_gn_param_shapes(embedding_dim).items():
        sh[f"encoder.resnet18.{k}"] = v
    for k, v in unet_param_shapes(input_dim, embedding_dim + additional_global_cond_dim, down_dims).items():
        sh[f"unet.{k}"] = v
    return sh


def blob_checksum(blob: np.ndarray):
    """Fletcher-style (s1, s2) over the blob's 32-bit words, both mod 2**64 (what ditree_load_weights re-computes)."""
    w = np.ascontiguousarray(blob, dtype=np.float32).view(np.uint32).reshape(-1)
    n = w.size
    s1 = s2 = 0
    mask = (1 << 64) - 1
    with np.errstate(over="ignore"):
        for lo in range(0, n, 1 << 24):
            c = w[lo:lo + (1 << 24)].astype(np.uint64)
            s1 = (s1 + int(c.sum(dtype=np.uint64))) & mask
            s2 = (s2 + int((c * np.arange(n - lo, n - lo - c.size, -1, dtype=np.uint64)).sum(dtype=np.uint64))) & mask
    return s1, s2


def pack_state_dict(state_dict, pred_horizon=None, local_map_size=None, checksum=True):
    """-> (blob float32 ndarray, manifest text) in the format ditree_load_weights parses:
    one line per tensor: ``name offset n_elems ndim d0 d1 ...``; ``#config`` carries what the tensor shapes do not
    tell (pred_horizon, local_map_size), ``#checksum`` guards the blob."""
    lines, chunks, off = [], [], 0
    for name, t in state_dict.items():
        a = t.detach().to("cpu", torch.float32).contiguous().numpy().reshape(-1)
        dims = " ".join(str(int(d)) for d in t.shape)
        lines.append(f"{name} {off} {a.size} {t.dim()} {dims}")
        chunks.append((off, a))
        off += a.size
    blob = np.empty(off, dtype=np.float32)        # filled slice by slice: np.concatenate of ~200 chunks took 15 x as long (measured)
    for o, a in chunks:
        blob[o:o + a.size] = a
    cfg = []
    if pred_horizon is not None:
        cfg.append(f"pred_horizon {int(pred_horizon)}")
    if local_map_size is not None:
        cfg.append(f"local_map_size {int(local_map_size)}")
    if cfg:
        lines.append("#config " + " ".join(cfg))
    if checksum:
        s1, s2 = blob_checksum(blob)
        lines.append(f"#checksum {s1:x} {s2:x}")
    return blob, "\n".join(lines) + "\n"
