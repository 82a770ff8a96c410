"""Tensor inventory of the SDv1.5 `AutoencoderKL` *encoder* half (+ `quant_conv`).

The reference turns an image into the latent `x` it scores with
`vae.encode(img).latent_dist.sample() * vae.config.scaling_factor`
(`diffmining/typicality/compute.py:91-93,137`, `diffmining/typicality/dift.py:187`).  The VAE lives in
the un-vendored diffusers dependency; this module restates the names and shapes of the encoder's
state dict from the public SDv1.5 `vae/config.json` (block_out_channels 128/256/512/512,
layers_per_block 2, 32 groups, latent_channels 4, scaling_factor 0.18215) so a checkpoint can be
checked before packing and synthetic weights of that architecture can be generated offline.
`decoder.*` / `post_quant_conv.*` tensors of a full VAE state dict are not on the path and are ignored.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215

    @property
    def downscale(self) -> int:
        return 2 ** (len(self.block_out_channels) - 1)


SD15_VAE = VAEConfig()

# pre-0.15 diffusers / original on-disk names of the mid-block attention -> current names
LEGACY_ATTN_NAMES = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _conv(name, cout, cin, k):
    return [(f"{name}.weight", (cout, cin, k, k)), (f"{name}.bias", (cout,))]


def _norm(name, c):
    return [(f"{name}.weight", (c,)), (f"{name}.bias", (c,))]


def _resnet(name, cin, cout):
    t = _norm(f"{name}.norm1", cin) + _conv(f"{name}.conv1", cout, cin, 3)
    t += _norm(f"{name}.norm2", cout) + _conv(f"{name}.conv2", cout, cout, 3)
    if cin != cout:
        t += _conv(f"{name}.conv_shortcut", cout, cin, 1)
    return t


def vae_encoder_tensor_spec(cfg: VAEConfig = SD15_VAE) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list: `encoder.*` and `quant_conv.*` of `AutoencoderKL.state_dict()`."""
    boc = cfg.block_out_channels
    t: List[Tuple[str, Tuple[int, ...]]] = []
    t += _conv("encoder.conv_in", boc[0], cfg.in_channels, 3)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            t += _resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(boc) - 1:
            t += _conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        cin = cout
    c = boc[-1]
    t += _resnet("encoder.mid_block.resnets.0", c, c)
    a = "encoder.mid_block.attentions.0"
    t += _norm(f"{a}.group_norm", c)
    for leaf in ("to_q", "to_k", "to_v", "to_out.0"):
        t += [(f"{a}.{leaf}.weight", (c, c)), (f"{a}.{leaf}.bias", (c,))]
    t += _resnet("encoder.mid_block.resnets.1", c, c)
    t += _norm("encoder.conv_norm_out", c)
    t += _conv("encoder.conv_out", 2 * cfg.latent_channels, c, 3)
    t += _conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return t


def vae_encoder_param_count(cfg: VAEConfig = SD15_VAE) -> int:
    n = 0
    for _, shp in vae_encoder_tensor_spec(cfg):
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


def canonical_vae_name(name: str):
    """Map a key of a VAE state dict to the name used here; None for tensors off the path."""
    if name.startswith("vae."):
        name = name[4:]
    if name.startswith("decoder.") or name.startswith("post_quant_conv."):
        return None
    if ".attentions.0." in name:
        head, leaf = name.split(".attentions.0.", 1)
        mod, _, wb = leaf.rpartition(".")
        if mod in LEGACY_ATTN_NAMES:
            return f"{head}.attentions.0.{LEGACY_ATTN_NAMES[mod]}.{wb}"
    return name
