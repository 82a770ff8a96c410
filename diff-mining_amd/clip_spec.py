"""Tensor inventory of the CLIP ViT-L/14 text tower (`CLIPTextModel`) the reference conditions on.

`CategoryFeatures.embed` (diffmining/typicality/compute.py:39-51) tokenises one prompt per category and
takes `self.clip(tokens)[0]` = `last_hidden_state` [n, 77, 768] of `openai/clip-vit-large-patch14-336`
(or `geolocal/StreetCLIP`, same architecture; compute.py:60-68).  The model lives in the `transformers`
dependency; this module restates the names and shapes of its state dict (123,060,480 parameters, 196
tensors) so a checkpoint can be checked before packing and synthetic weights can be generated offline.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5          # hidden_act = quick_gelu: x * sigmoid(1.702 x)
    bos_token_id: int = 49406
    eos_token_id: int = 49407


CLIP_L14_TEXT = CLIPTextConfig()


def clip_text_tensor_spec(cfg: CLIPTextConfig = CLIP_L14_TEXT) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list; names are relative to `text_model.` (see `canonical_clip_name`)."""
    h, f = cfg.hidden_size, cfg.intermediate_size
    t: List[Tuple[str, Tuple[int, ...]]] = [
        ("embeddings.token_embedding.weight", (cfg.vocab_size, h)),
        ("embeddings.position_embedding.weight", (cfg.max_position_embeddings, h)),
    ]
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            t += [(f"{p}.self_attn.{proj}.weight", (h, h)), (f"{p}.self_attn.{proj}.bias", (h,))]
        t += [(f"{p}.layer_norm1.weight", (h,)), (f"{p}.layer_norm1.bias", (h,))]
        t += [(f"{p}.mlp.fc1.weight", (f, h)), (f"{p}.mlp.fc1.bias", (f,))]
        t += [(f"{p}.mlp.fc2.weight", (h, f)), (f"{p}.mlp.fc2.bias", (h,))]
        t += [(f"{p}.layer_norm2.weight", (h,)), (f"{p}.layer_norm2.bias", (h,))]
    t += [("final_layer_norm.weight", (h,)), ("final_layer_norm.bias", (h,))]
    return t


def clip_text_param_count(cfg: CLIPTextConfig = CLIP_L14_TEXT) -> int:
    n = 0
    for _, shp in clip_text_tensor_spec(cfg):
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


def canonical_clip_name(name: str):
    """Key of a `CLIPTextModel` / pipeline state dict -> name used here; None for buffers off the path."""
    for pre in ("text_encoder.", "text_model."):
        if name.startswith(pre):
            name = name[len(pre):]
    if name.startswith("text_model."):
        name = name[len("text_model."):]
    if name.endswith("position_ids"):
        return None
    return name
