"""ORACLE — test infrastructure only.  Pinned against `transformers.CLIPTextModel` (see below).

CPU restatement (PyTorch fp32, `torch.nn.functional` primitives) of the CLIP ViT-L/14 text tower whose
`last_hidden_state` the reference uses as prompt embedding `c` (`CategoryFeatures.embed`,
diffmining/typicality/compute.py:39-51: `self.clip(tokens.to(device))[0].float()`).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

The arithmetic lives in the third-party `transformers` (reference pin 4.36.0, `environment.yaml`).  That
package IS importable in the build container (5.x; the CLIP text model is unchanged), so unlike the
U-Net / VAE oracles this one is pinned: `tests/make_golden.py` instantiates `transformers.CLIPTextModel`
with the synthetic weights, runs it on synthetic token ids and commits inputs + `last_hidden_state` as
`tests/golden/clip_text.npz`; `tests/test_oracle.py` checks this restatement against that fixture
(fp32, 1e-5).  transformers itself never ships to the GPU box.

Restated forward (`CLIPTextTransformer.forward`): token + position embedding; 12 pre-LN layers:
x += out_proj(softmax(q k^T * d^-0.5 + causal mask) v) with q/k/v = Linear(LN1(x)) split in 12 heads of
64; x += fc2(quick_gelu(fc1(LN2(x)))), quick_gelu(y) = y * sigmoid(1.702 y); final LayerNorm.
`autocast=True` emulates the fp16 model of compute.py:68 (`torch_dtype=torch.float16`): every op emits
fp16 (LayerNorm / softmax accumulate in fp32 internally).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .unet_ref import _SD, _r

HEADS = 12
EPS = 1e-5
LAYERS = 12


def _linear(p: _SD, name, x, ac):
    return _r(F.linear(_r(x, ac), p(name + ".weight"), p(name + ".bias")), ac)


def _ln(p: _SD, name, x, ac):
    return _r(F.layer_norm(x, (x.shape[-1],), p(name + ".weight"), p(name + ".bias"), EPS), ac)


def clip_text_forward(sd: Dict[str, torch.Tensor], input_ids: torch.Tensor, autocast: bool = False,
                      used_keys=None) -> torch.Tensor:
    """input_ids [n, T<=77] int64 -> last_hidden_state [n, T, 768] fp32."""
    p = _SD(sd)
    ac = autocast
    n, T = input_ids.shape
    x = _r(p("embeddings.token_embedding.weight")[input_ids] + p("embeddings.position_embedding.weight")[:T][None], ac)
    mask = torch.full((T, T), float("-inf")).triu(1)
    for i in range(LAYERS):
        b = f"encoder.layers.{i}"
        h = _ln(p, b + ".layer_norm1", x, ac)
        q = _linear(p, b + ".self_attn.q_proj", h, ac)
        k = _linear(p, b + ".self_attn.k_proj", h, ac)
        v = _linear(p, b + ".self_attn.v_proj", h, ac)
        d = q.shape[-1] // HEADS
        q = _r(q * d ** -0.5, ac)
        qh, kh, vh = (t.view(n, T, HEADS, d).transpose(1, 2) for t in (q, k, v))
        s = torch.matmul(qh, kh.transpose(-1, -2)) + mask
        a = _r(torch.matmul(_r(torch.softmax(s, dim=-1), ac), vh), ac)
        a = a.transpose(1, 2).reshape(n, T, HEADS * d)
        x = _r(x + _linear(p, b + ".self_attn.out_proj", a, ac), ac)
        h = _ln(p, b + ".layer_norm2", x, ac)
        h = _linear(p, b + ".mlp.fc1", h, ac)
        h = _r(h * torch.sigmoid(1.702 * h), ac)
        x = _r(x + _linear(p, b + ".mlp.fc2", h, ac), ac)
    x = _ln(p, "final_layer_norm", x, ac)
    if used_keys is not None:
        used_keys.update(p.used)
    return x


def category_prompts(which: str, categories):
    """The prompt templates of `CategoryFeatures.embed` (compute.py:41-48); '' is the null prompt."""
    if which == "faces":
        return [(f"Portrait at the {c}'s." if len(c) else "Portrait.") for c in categories]
    if which == "cars":
        return [(f"A car at the {c}'s." if len(c) else "A car.") for c in categories]
    if which == "places":
        return [("Image of " + c.replace("_", " ") + "." if len(c) else "") for c in categories]
    return [(f"{c}" if len(c) else "") for c in categories]
