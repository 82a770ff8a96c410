"""Tensor inventory of the SDv1.5 `UNet2DConditionModel` state dict.

The typicality hot path (reference `diffmining/typicality/compute.py:100`,
`diffmining/typicality/dift.py:191`) calls a diffusers-0.24 U-Net whose weights
arrive as a diffusers-named state dict (`unet/diffusion_pytorch_model.safetensors`).
This module restates the *names and shapes* of that state dict from the public
SDv1.5 `unet/config.json` so that

  * the engine can check a checkpoint before packing it (686 tensors,
    859,520,964 parameters for the stock config), and
  * synthetic weights of exactly that architecture can be generated when no
    checkpoint is available (there is none in the build environment).

Nothing here is arithmetic; see `engine.py` for the HIP path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    """Subset of diffusers' `UNet2DConditionModel` config the hot path depends on."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    num_heads: int = 8                 # `attention_head_dim: 8` is the HEAD COUNT in SDv1.5
    norm_num_groups: int = 32
    norm_eps: float = 1e-5             # ResNet GroupNorm / conv_norm_out
    attn_norm_eps: float = 1e-6        # Transformer2DModel.norm
    ln_eps: float = 1e-5
    # which down blocks carry cross attention (CrossAttnDownBlock2D ×3, DownBlock2D)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    # up blocks: UpBlock2D, CrossAttnUpBlock2D ×3
    up_has_attn: Tuple[bool, ...] = (False, True, True, True)
    context_len: int = 77
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def n_blocks(self) -> int:
        return len(self.block_out_channels)


SD15 = UNetConfig()


def _conv(name, cout, cin, k):
    return [(f"{name}.weight", (cout, cin, k, k)), (f"{name}.bias", (cout,))]


def _linear(name, cout, cin, bias=True):
    out = [(f"{name}.weight", (cout, cin))]
    if bias:
        out.append((f"{name}.bias", (cout,)))
    return out


def _norm(name, c):
    return [(f"{name}.weight", (c,)), (f"{name}.bias", (c,))]


def _resnet(name, cin, cout, temb):
    t = []
    t += _norm(f"{name}.norm1", cin)
    t += _conv(f"{name}.conv1", cout, cin, 3)
    t += _linear(f"{name}.time_emb_proj", cout, temb)
    t += _norm(f"{name}.norm2", cout)
    t += _conv(f"{name}.conv2", cout, cout, 3)
    if cin != cout:
        t += _conv(f"{name}.conv_shortcut", cout, cin, 1)
    return t


def _transformer(name, c, ctx):
    t = []
    t += _norm(f"{name}.norm", c)
    t += _conv(f"{name}.proj_in", c, c, 1)
    b = f"{name}.transformer_blocks.0"
    t += _norm(f"{b}.norm1", c)
    t += _linear(f"{b}.attn1.to_q", c, c, bias=False)
    t += _linear(f"{b}.attn1.to_k", c, c, bias=False)
    t += _linear(f"{b}.attn1.to_v", c, c, bias=False)
    t += _linear(f"{b}.attn1.to_out.0", c, c)
    t += _norm(f"{b}.norm2", c)
    t += _linear(f"{b}.attn2.to_q", c, c, bias=False)
    t += _linear(f"{b}.attn2.to_k", c, ctx, bias=False)
    t += _linear(f"{b}.attn2.to_v", c, ctx, bias=False)
    t += _linear(f"{b}.attn2.to_out.0", c, c)
    t += _norm(f"{b}.norm3", c)
    t += _linear(f"{b}.ff.net.0.proj", 8 * c, c)
    t += _linear(f"{b}.ff.net.2", c, 4 * c)
    t += _conv(f"{name}.proj_out", c, c, 1)
    return t


def up_block_channels(cfg: UNetConfig):
    """(in_channels, skip_channels, out_channels) of every up-block ResNet, in order.

    Mirrors the skip-stack bookkeeping of the diffusers U-Net the reference drives
    (`dift.py:133-165` walks the same `up_blocks`): skips are popped three at a time.
    """
    boc = list(cfg.block_out_channels)
    rev = boc[::-1]
    out = []
    prev = rev[0]
    for i in range(cfg.n_blocks):
        o = rev[i]
        inp = rev[min(i + 1, cfg.n_blocks - 1)]
        blk = []
        for j in range(cfg.layers_per_block + 1):
            skip = inp if j == cfg.layers_per_block else o
            rin = prev if j == 0 else o
            blk.append((rin, skip, o))
        out.append(blk)
        prev = o
    return out


def unet_tensor_spec(cfg: UNetConfig = SD15) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list of the diffusers state dict for `cfg`."""
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    t: List[Tuple[str, Tuple[int, ...]]] = []
    t += _conv("conv_in", boc[0], cfg.in_channels, 3)
    t += _linear("time_embedding.linear_1", temb, boc[0])
    t += _linear("time_embedding.linear_2", temb, temb)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            t += _resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb)
            if cfg.down_has_attn[i]:
                t += _transformer(f"down_blocks.{i}.attentions.{j}", cout, ctx)
        if i != cfg.n_blocks - 1:
            t += _conv(f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        cin = cout
    c = boc[-1]
    t += _resnet("mid_block.resnets.0", c, c, temb)
    t += _transformer("mid_block.attentions.0", c, ctx)
    t += _resnet("mid_block.resnets.1", c, c, temb)
    for i, blk in enumerate(up_block_channels(cfg)):
        for j, (rin, skip, o) in enumerate(blk):
            t += _resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, o, temb)
            if cfg.up_has_attn[i]:
                t += _transformer(f"up_blocks.{i}.attentions.{j}", o, ctx)
        if i != cfg.n_blocks - 1:
            t += _conv(f"up_blocks.{i}.upsamplers.0.conv", o, o, 3)
    t += _norm("conv_norm_out", boc[0])
    t += _conv("conv_out", cfg.out_channels, boc[0], 3)
    return t


def param_count(cfg: UNetConfig = SD15) -> int:
    n = 0
    for _, shp in unet_tensor_spec(cfg):
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


# ---- unet/config.json of an exported pipeline directory ---------------------------------------------------------
# The reference loads `StableDiffusionPipeline.from_pretrained(model_path)` (compute.py:65-70), i.e. diffusers builds the
# U-Net that `<model_path>/unet/config.json` describes (written by `--export-only`, finetuning/base.py:245-250).  The
# engine implements exactly one architecture, so a config that describes another one is rejected BY KEY before any
# tensor is packed (otherwise the first symptom would be a tensor-shape message, or — for options that do not change a
# shape, like `use_linear_projection` with equal sizes or `upcast_attention` — silently different arithmetic).
SD15_UNET_CONFIG = {
    "in_channels": 4, "out_channels": 4, "block_out_channels": [320, 640, 1280, 1280], "layers_per_block": 2,
    "cross_attention_dim": 768, "attention_head_dim": 8, "norm_num_groups": 32, "norm_eps": 1e-5, "act_fn": "silu",
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "mid_block_type": "UNetMidBlock2DCrossAttn", "flip_sin_to_cos": True, "freq_shift": 0, "downsample_padding": 1,
    "mid_block_scale_factor": 1, "center_input_sample": False,
}
# keys later diffusers versions add; absent (SDv1.5's own config.json) or at these defaults is fine, anything else is not
_OPTIONAL_DEFAULTS = {
    "use_linear_projection": False, "dual_cross_attention": False, "only_cross_attention": False,
    "upcast_attention": False, "resnet_time_scale_shift": "default", "time_embedding_type": "positional",
    "class_embed_type": None, "num_class_embeds": None, "addition_embed_type": None, "encoder_hid_dim": None,
    "encoder_hid_dim_type": None, "transformer_layers_per_block": 1, "num_attention_heads": None,
    "time_embedding_dim": None, "time_embedding_act_fn": None, "timestep_post_act": None, "time_cond_proj_dim": None,
    "conv_in_kernel": 3, "conv_out_kernel": 3, "projection_class_embeddings_input_dim": None,
    "class_embeddings_concat": False, "mid_block_only_cross_attention": None, "cross_attention_norm": None,
    "resnet_skip_time_act": False, "resnet_out_scale_factor": 1.0, "attention_type": "default", "dropout": 0.0,
    "addition_time_embed_dim": None, "addition_embed_type_num_heads": 64, "reverse_transformer_layers_per_block": None,
}


def check_unet_config(cfg: dict) -> None:
    """Raise ValueError naming every key of a `unet/config.json` that differs from the SDv1.5 architecture."""
    bad = []
    cls = cfg.get("_class_name")
    if cls is not None and cls != "UNet2DConditionModel":
        bad.append(f"_class_name = {cls!r} (want 'UNet2DConditionModel')")

    def same(a, b):
        if isinstance(a, (list, tuple)) or isinstance(b, (list, tuple)):
            return isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)) and list(a) == list(b)
        if isinstance(a, float) or isinstance(b, float):
            return a is not None and b is not None and abs(float(a) - float(b)) <= 1e-12 + 1e-6 * abs(float(b))
        return a == b
    for k, want in SD15_UNET_CONFIG.items():
        if k not in cfg:
            if k in ("mid_block_type", "center_input_sample", "mid_block_scale_factor"):      # newer / optional keys
                continue
            bad.append(f"{k} missing (want {want!r})")
        elif not same(cfg[k], want):
            bad.append(f"{k} = {cfg[k]!r} (want {want!r})")
    for k, default in _OPTIONAL_DEFAULTS.items():
        if k in cfg and cfg[k] is not None and not same(cfg[k], default):
            if k == "num_attention_heads" and same(cfg[k], 8):
                continue
            bad.append(f"{k} = {cfg[k]!r} (only {default!r} is implemented)")
    if bad:
        raise ValueError("unet/config.json does not describe the SDv1.5 U-Net this engine implements: " + "; ".join(bad))


def check_unet_config_file(path: str) -> dict:
    import json
    with open(path) as f:
        cfg = json.load(f)
    check_unet_config(cfg)
    return cfg
