"""ORACLE — test infrastructure only.  PARITY UNPINNED.

CPU restatement (PyTorch fp32, `torch.nn.functional` primitives only) of the SDv1.5
`AutoencoderKL.encode(...).latent_dist.sample() * scaling_factor` step the reference runs before
scoring (`diffmining/typicality/compute.py:91-93,137`; `diffmining/typicality/dift.py:187`;
SURVEY.md §8a R7, §8f rank 2).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this module.

The arithmetic lives in the un-vendored `diffusers==0.24.0` (`environment.yaml:15`); diffusers and
the weights are absent here and the reference ships no tests, so this file restates the public
SDv1.5 `vae/config.json` architecture and diffusers-0.24 semantics:

  Encoder: conv_in 3->128 (3x3, pad 1); 4 x DownEncoderBlock2D (128, 256, 512, 512; two
  ResnetBlock2D each, GroupNorm(32, eps 1e-6) - SiLU - conv3x3, no time embedding, 1x1
  `conv_shortcut` when channels change; blocks 0-2 end in Downsample2D(padding=0):
  F.pad(x, (0,1,0,1)) then conv3x3 stride 2); UNetMidBlock2D: resnet, single-head attention
  (GroupNorm(32, 1e-6), to_q/to_k/to_v/to_out.0 Linear 512 with bias, head_dim 512, residual),
  resnet; GroupNorm(32, 1e-6) - SiLU - conv_out 512->8; quant_conv 1x1 8->8.
  DiagonalGaussianDistribution: mean, logvar = chunk(moments, 2, dim=1); logvar clamped to
  [-30, 20]; std = exp(0.5 logvar); sample = mean + std * noise.

It is pinned only structurally (34,163,592 encoder parameters + 72 of quant_conv, 108 tensors,
shape walk, algebraic properties) in `tests/test_oracle.py`.

`autocast=True` emulates `@torch.autocast('cuda')` of `encode_vae` (compute.py:91) with the fp16
pipeline weights (compute.py:65-70): conv / linear / attention emit fp16, group_norm, softmax and exp
run in fp32.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .unet_ref import _SD, _r

GROUPS = 32
EPS = 1e-6
BLOCK_OUT = (128, 256, 512, 512)
SCALING_FACTOR = 0.18215


def _conv(p: _SD, name, x, ac, stride=1, padding=1):
    return _r(F.conv2d(_r(x, ac), p(name + ".weight"), p(name + ".bias"), stride=stride, padding=padding), ac)


def _linear(p: _SD, name, x, ac):
    return _r(F.linear(_r(x, ac), p(name + ".weight"), p(name + ".bias")), ac)


def _gn(p: _SD, name, x):
    return F.group_norm(x, GROUPS, p(name + ".weight"), p(name + ".bias"), EPS)


def _resnet(p: _SD, name, x, ac):
    """`ResnetBlock2D(temb_channels=None)`: GN-SiLU-conv1-GN-SiLU-conv2 + shortcut."""
    h = _conv(p, name + ".conv1", F.silu(_gn(p, name + ".norm1", x)), ac)
    h = _conv(p, name + ".conv2", F.silu(_gn(p, name + ".norm2", h)), ac)
    sc = _conv(p, name + ".conv_shortcut", x, ac, padding=0) if p.has(name + ".conv_shortcut.weight") else x
    return _r(sc + h, ac)


def _attention(p: _SD, name, x, ac):
    """`Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups=32)` on [B,C,H,W]."""
    B, C, H, W = x.shape
    h = _gn(p, name + ".group_norm", x.reshape(B, C, H * W)).transpose(1, 2)       # [B, HW, C]
    q = _linear(p, name + ".to_q", h, ac)
    k = _linear(p, name + ".to_k", h, ac)
    v = _linear(p, name + ".to_v", h, ac)
    s = torch.matmul(q, k.transpose(1, 2)) * (C ** -0.5)                             # fp32 scores
    o = _r(torch.matmul(_r(torch.softmax(s, dim=-1), ac), v), ac)
    o = _linear(p, name + ".to_out.0", o, ac)
    return _r(o.transpose(1, 2).reshape(B, C, H, W) + x, ac)


def vae_moments(sd: Dict[str, torch.Tensor], image: torch.Tensor, autocast: bool = True,
                used_keys: Optional[set] = None) -> torch.Tensor:
    """`quant_conv(encoder(image))` -> [B, 8, H/8, W/8] (mean | logvar), `AutoencoderKL.encode`."""
    p = _SD(sd)
    ac = autocast
    h = _conv(p, "encoder.conv_in", image.float(), ac)
    for i in range(len(BLOCK_OUT)):
        for j in range(2):
            h = _resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, ac)
        if i != len(BLOCK_OUT) - 1:
            h = F.pad(h, (0, 1, 0, 1))                                               # Downsample2D(padding=0)
            h = _conv(p, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, ac, stride=2, padding=0)
    h = _resnet(p, "encoder.mid_block.resnets.0", h, ac)
    h = _attention(p, "encoder.mid_block.attentions.0", h, ac)
    h = _resnet(p, "encoder.mid_block.resnets.1", h, ac)
    h = _conv(p, "encoder.conv_out", F.silu(_gn(p, "encoder.conv_norm_out", h)), ac)
    m = _conv(p, "quant_conv", h, ac, padding=0)
    if used_keys is not None:
        used_keys.update(p.used)
    return m


def posterior_sample(moments: torch.Tensor, noise: Optional[torch.Tensor],
                     scaling_factor: float = SCALING_FACTOR) -> torch.Tensor:
    """`DiagonalGaussianDistribution(moments).sample() * scaling_factor` (compute.py:93) with the draw
    injected (`noise=None` -> the posterior mode).  exp and the affine run in fp32 (autocast promotes exp)."""
    mean, logvar = torch.chunk(moments.float(), 2, dim=1)
    if noise is None:
        return mean * scaling_factor
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return (mean + std * noise.float()) * scaling_factor


def vae_encode(sd, image, noise=None, autocast=True, scaling_factor=SCALING_FACTOR) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (latents [B,4,h,w] fp32, moments [B,8,h,w] fp32)."""
    m = vae_moments(sd, image, autocast)
    return posterior_sample(m, noise, scaling_factor), m
