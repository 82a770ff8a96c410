"""ORACLE — test infrastructure only.  PARITY UNPINNED.

CPU restatement (PyTorch fp32, `torch.nn.functional` primitives only) of the arithmetic on
diff-mining's typicality hot path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product path
(`diff-mining_amd/`) never does and fails loudly when its HIP library is missing.

Why "parity unpinned": the arithmetic of this path is NOT in the reference repository.  It lives
in the un-vendored dependency `diffusers==0.24.0` (`/root/reference/environment.yaml:15`), reached
from `diffmining/typicality/compute.py:99-101` (`scheduler.add_noise`, `unet(...)`, `F.mse_loss`)
and `diffmining/typicality/dift.py:24-169`.  diffusers, torchvision and the SD weights are absent
from the build container and the reference ships no tests or golden vectors,
so this file restates the *published* SDv1.5 architecture (`unet/config.json`) and diffusers-0.24
semantics, and is pinned only structurally: 859,520,964 parameters / 686 diffusers-named tensors,
scheduler-table and sinusoid known answers, shape walks and algebraic properties
(`tests/test_oracle.py`).

Each function cites the reference call site it stands in for.

Pinned parts (r04): the CONTROL FLOW restated here — `draw_noise_and_timesteps`, `compute_loss`, `compute_losses` — and the grid consumers
(`load_typicality`, `load_typicality_norm`, `d_compute`, `normalize_map`) reproduce, bit for bit, what the reference's own classes /
functions produce when they are run from /root/reference with this module's U-Net plugged in as `pipe.unet`
(tests/make_golden_host.py, tests/make_golden_consumers.py -> tests/golden/host_ref.npz, consumers_ref.npz).  The U-Net / scheduler
ARITHMETIC (the diffusers part) is what remains unpinned.

Two numeric modes:
  * `autocast=False` — plain fp32 everywhere (the mathematical ground truth).
  * `autocast=True`  — emulates `torch.autocast('cuda', dtype=float16)` as used at
    `compute.py:98`: conv / linear / attention take fp16 inputs and emit fp16 (fp32 accumulate),
    group_norm / layer_norm / softmax / mse_loss run in fp32, the residual stream is fp16.
    Emulated by rounding through fp16 at exactly those points while computing in fp32 on CPU.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# configuration (kept local so that the oracle imports nothing from the product package)
# --------------------------------------------------------------------------------------------
class RefConfig:
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, cross_attention_dim=768, num_heads=8, norm_num_groups=32,
                 norm_eps=1e-5, attn_norm_eps=1e-6, ln_eps=1e-5,
                 down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True)):
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.block_out_channels = tuple(block_out_channels)
        self.layers_per_block = layers_per_block
        self.cross_attention_dim = cross_attention_dim
        self.num_heads = num_heads
        self.norm_num_groups = norm_num_groups
        self.norm_eps = norm_eps
        self.attn_norm_eps = attn_norm_eps
        self.ln_eps = ln_eps
        self.down_has_attn = tuple(down_has_attn)
        self.up_has_attn = tuple(up_has_attn)


SD15_REF = RefConfig()


class _SD:
    """State-dict accessor that records which keys the forward consumed (structural test)."""

    def __init__(self, sd: Dict[str, torch.Tensor]):
        self.sd = sd
        self.used = set()

    def __call__(self, name: str) -> torch.Tensor:
        self.used.add(name)
        return self.sd[name]

    def has(self, name: str) -> bool:
        return name in self.sd


# Rounding-point ablation (`python tools/oracle_noise.py ablation` -> profiles/r04_oracle_rounding_ablation.txt; no test covers the site tags): every fp16 rounding of
# the autocast emulation carries a site tag; a tag in ROUND_OFF is skipped (that value stays fp32), so the contribution of one
# class of roundings to the autocast-vs-fp32 distance can be named.  Empty by default = the emulation described above.
#   "in"     operand of a conv / linear (what autocast casts to fp16 before the op)       "out"   result of a conv / linear
#   "res"    a residual sum (ResNet output, the three transformer residuals, proj_out + input)     "temb"  conv1 + time embedding
#   "p"      softmax output fed to P.V     "o"  attention output     "geglu"  GELU(gate) and value * gate
#   "emb"    the time-embedding MLP chain (sinusoid, SiLU outputs)     "x"  the rounding of the sample / prompt at the boundary
ROUND_OFF = set()


def _r(x: torch.Tensor, autocast: bool, site: str = "out") -> torch.Tensor:
    """Round through fp16 when emulating autocast (output of an fp16 op)."""
    return x.half().float() if (autocast and site not in ROUND_OFF) else x


# --------------------------------------------------------------------------------------------
# scheduler — diffusers `SchedulerMixin.add_noise` as called at compute.py:99 / dift.py:190
# --------------------------------------------------------------------------------------------
def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    """`scaled_linear` betas of the SDv1.5 scheduler config (PNDM/DDPM/DDIM share it)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
              acp: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`scheduler.add_noise(x, noise, t)` (compute.py:99).  The table is cast to x.dtype FIRST,
    then sqrt / products / sum are all taken in that dtype."""
    if acp is None:
        acp = alphas_cumprod()
    a = acp.to(dtype=x.dtype)[timesteps]
    sa = (a ** 0.5).flatten()
    sb = ((1 - a) ** 0.5).flatten()
    while sa.dim() < x.dim():
        sa = sa.unsqueeze(-1)
        sb = sb.unsqueeze(-1)
    return sa * x + sb * noise


# --------------------------------------------------------------------------------------------
# U-Net pieces (diffusers 0.24 `UNet2DConditionModel`, restated)
# --------------------------------------------------------------------------------------------
def timestep_sinusoid(timesteps: torch.Tensor, dim: int = 320) -> torch.Tensor:
    """`Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)` -> [B, dim] fp32 = [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _conv(p: _SD, name: str, x, ac, stride=1, padding=1):
    return _r(F.conv2d(_r(x, ac, "in"), p(name + ".weight"), p(name + ".bias"), stride=stride, padding=padding), ac, "out")


def _linear(p: _SD, name: str, x, ac, bias=True):
    return _r(F.linear(_r(x, ac, "in"), p(name + ".weight"), p(name + ".bias") if bias else None), ac, "out")


# Optional observer of every normalisation input: PROBE(kind, name, x) with kind "gn" / "ln" (tests: |mean| / std and max |x|
# of the operating point a weight set puts the norms at).  None = off.
PROBE = None


def _gn(p: _SD, name: str, x, groups, eps):
    # autocast promotes group_norm to fp32: the output is NOT rounded to fp16
    if PROBE is not None:
        PROBE("gn", name, x)
    return F.group_norm(x, groups, p(name + ".weight"), p(name + ".bias"), eps)


def _ln(p: _SD, name: str, x, eps):
    if PROBE is not None:
        PROBE("ln", name, x)
    return F.layer_norm(x, (x.shape[-1],), p(name + ".weight"), p(name + ".bias"), eps)


def _resnet(p: _SD, name: str, x, temb_act, cfg: RefConfig, ac):
    """`ResnetBlock2D`: GN-SiLU-conv1 (+ time_emb_proj(SiLU(temb))) -GN-SiLU-conv2, + shortcut."""
    h = F.silu(_gn(p, name + ".norm1", x, cfg.norm_num_groups, cfg.norm_eps))
    h = _conv(p, name + ".conv1", h, ac)
    t = _linear(p, name + ".time_emb_proj", temb_act, ac)
    h = _r(h + t[:, :, None, None], ac, "temb")
    h = F.silu(_gn(p, name + ".norm2", h, cfg.norm_num_groups, cfg.norm_eps))
    h = _conv(p, name + ".conv2", h, ac)
    if p.has(name + ".conv_shortcut.weight"):
        x = _conv(p, name + ".conv_shortcut", x, ac, padding=0)
    return _r(x + h, ac, "res")


def _attention(p: _SD, name: str, x, ctx, heads, ac):
    """`Attention` with `AttnProcessor2_0` (SDPA), scale = head_dim**-0.5, no mask."""
    B, T, C = x.shape
    kv = x if ctx is None else ctx
    q = _linear(p, name + ".to_q", x, ac, bias=False)
    k = _linear(p, name + ".to_k", kv, ac, bias=False)
    v = _linear(p, name + ".to_v", kv, ac, bias=False)
    d = C // heads
    q = q.view(B, T, heads, d).transpose(1, 2)
    k = k.view(B, kv.shape[1], heads, d).transpose(1, 2)
    v = v.view(B, kv.shape[1], heads, d).transpose(1, 2)
    kT = k.transpose(-1, -2)

    def rows(qb):
        s = torch.matmul(qb, kT) * (d ** -0.5)
        pr = torch.softmax(s, dim=-1)            # fp32 softmax on fp32 scores (flash kernels)
        return _r(torch.matmul(_r(pr, ac, "p"), v), ac, "o")  # P is fed to the PV matmul in fp16
    # the softmax is per query row, so blocks of query rows are independent: at 128 x 128 latents (16 384 tokens, the
    # X-ray case) the full score tensor would be 8.6 GB per sample — keep it below 2^28 elements at a time
    qc = max(1, (1 << 28) // max(1, B * heads * kv.shape[1]))
    o = rows(q) if T <= qc else torch.cat([rows(q[:, :, i:i + qc]) for i in range(0, T, qc)], dim=2)
    o = o.transpose(1, 2).reshape(B, T, C)
    return _linear(p, name + ".to_out.0", o, ac)


def _transformer(p: _SD, name: str, x, ctx, cfg: RefConfig, ac):
    """`Transformer2DModel` (conv proj_in/out) around one `BasicTransformerBlock`."""
    B, C, H, W = x.shape
    res = x
    h = _gn(p, name + ".norm", x, cfg.norm_num_groups, cfg.attn_norm_eps)
    h = _conv(p, name + ".proj_in", h, ac, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = name + ".transformer_blocks.0"
    h = _r(_attention(p, b + ".attn1", _ln(p, b + ".norm1", h, cfg.ln_eps), None, cfg.num_heads, ac) + h, ac, "res")
    h = _r(_attention(p, b + ".attn2", _ln(p, b + ".norm2", h, cfg.ln_eps), ctx, cfg.num_heads, ac) + h, ac, "res")
    n3 = _ln(p, b + ".norm3", h, cfg.ln_eps)
    proj = _linear(p, b + ".ff.net.0.proj", n3, ac)
    a, g = proj.chunk(2, dim=-1)
    ff = _r(a * _r(F.gelu(g), ac, "geglu"), ac, "geglu")     # GEGLU, erf GELU in fp16 under autocast
    h = _r(_linear(p, b + ".ff.net.2", ff, ac) + h, ac, "res")
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = _conv(p, name + ".proj_out", h, ac, padding=0)
    return _r(h + res, ac, "res")


def unet_forward(sd: Dict[str, torch.Tensor], sample: torch.Tensor, timesteps: torch.Tensor,
                 encoder_hidden_states: torch.Tensor, cfg: RefConfig = SD15_REF, autocast: bool = False,
                 up_ft_indices: Optional[Sequence[int]] = None, used_keys: Optional[set] = None):
    """`UNet2DConditionModel.forward(sample, t, ctx).sample`  (compute.py:100), or — when
    `up_ft_indices` is given — `MyUNet2DConditionModel.forward` (dift.py:24-169): early exit after
    `up_blocks[max(up_ft_indices)]`, returning {'up_ft': {i: feature}}.

    sample [B,4,h,w]; timesteps [B] or 0-dim int64; encoder_hidden_states [B,77,768].
    """
    p = _SD(sd)
    ac = autocast
    B = sample.shape[0]
    if timesteps.dim() == 0:
        timesteps = timesteps[None]
    timesteps = timesteps.expand(B)
    sample = _r(sample.float(), ac, "x")
    ctx = _r(encoder_hidden_states.float(), ac, "x")
    boc = cfg.block_out_channels
    nb = len(boc)
    n_up = nb - 1                                    # number of upsamplers
    fwd_up_size = any(s % (2 ** n_up) != 0 for s in sample.shape[-2:])   # dift.py:54-56

    # 1. time (dift.py:84-91)
    t_emb = _r(timestep_sinusoid(timesteps, boc[0]), ac, "emb")
    emb = _linear(p, "time_embedding.linear_1", t_emb, ac)
    emb = _r(F.silu(emb), ac, "emb")
    emb = _linear(p, "time_embedding.linear_2", emb, ac)
    temb_act = _r(F.silu(emb), ac, "emb")            # every ResNet applies SiLU before time_emb_proj

    # 2. conv_in
    h = _conv(p, "conv_in", sample, ac)
    skips = [h]
    # 3. down
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = _resnet(p, f"down_blocks.{i}.resnets.{j}", h, temb_act, cfg, ac)
            if cfg.down_has_attn[i]:
                h = _transformer(p, f"down_blocks.{i}.attentions.{j}", h, ctx, cfg, ac)
            skips.append(h)
        if i != nb - 1:
            h = _conv(p, f"down_blocks.{i}.downsamplers.0.conv", h, ac, stride=2, padding=1)
            skips.append(h)
    # 4. mid
    h = _resnet(p, "mid_block.resnets.0", h, temb_act, cfg, ac)
    h = _transformer(p, "mid_block.attentions.0", h, ctx, cfg, ac)
    h = _resnet(p, "mid_block.resnets.1", h, temb_act, cfg, ac)
    # 5. up
    up_ft = {}
    for i in range(nb):
        if up_ft_indices is not None and i > max(up_ft_indices):
            break
        n_res = cfg.layers_per_block + 1
        res_samples = skips[-n_res:]
        skips = skips[:-n_res]
        is_final = i == nb - 1
        up_size = skips[-1].shape[2:] if (not is_final and fwd_up_size) else None
        for j in range(n_res):
            h = torch.cat([h, res_samples[-1 - j]], dim=1)
            h = _resnet(p, f"up_blocks.{i}.resnets.{j}", h, temb_act, cfg, ac)
            if cfg.up_has_attn[i]:
                h = _transformer(p, f"up_blocks.{i}.attentions.{j}", h, ctx, cfg, ac)
        if not is_final:
            if up_size is None:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            else:
                h = F.interpolate(h, size=tuple(up_size), mode="nearest")
            h = _conv(p, f"up_blocks.{i}.upsamplers.0.conv", h, ac)
        if up_ft_indices is not None and i in up_ft_indices:
            up_ft[i] = h
    if used_keys is not None:
        used_keys |= p.used
    if up_ft_indices is not None:
        return {"up_ft": up_ft}
    # 6. post-process
    h = F.silu(_gn(p, "conv_norm_out", h, cfg.norm_num_groups, cfg.norm_eps))
    h = _conv(p, "conv_out", h, ac)
    if used_keys is not None:
        used_keys |= p.used
    return h


# --------------------------------------------------------------------------------------------
# scoring surface — diffmining/typicality/compute.py:95-160
# --------------------------------------------------------------------------------------------
def compute_loss(sd, x, noise, timesteps, c, cfg: RefConfig = SD15_REF, autocast: bool = True,
                 acp: Optional[torch.Tensor] = None, latent_dtype: torch.dtype = torch.float32):
    """`SD.compute_loss` (compute.py:95-102): add_noise -> U-Net -> per-element squared error.

    x [1 or 2B,4,h,w]; noise [2B,4,h,w]; timesteps [2B] int64; c [2B,77,768].  Returns loss [2B,4,h,w] fp32.

    `latent_dtype` (only meaningful with autocast=True) is the dtype x and noise arrive in:
      * torch.float32 — the reference's actual flow.  `encode_vae` runs under `@torch.autocast` (compute.py:91-93);
        `DiagonalGaussianDistribution` takes `std = exp(0.5 * logvar)` and autocast promotes `exp` to fp32, so
        `mean + std * sample` and hence x are fp32; `randn_like(x)` (:116) is fp32; `add_noise` (:99) casts the
        table to x.dtype = fp32 and stays fp32; autocast rounds the noisy latent to fp16 only as conv_in's
        input; `mse_loss(noise_pred.float(), noise)` (:101) sees the unrounded fp32 noise.
      * torch.float16 — an fp16 latent handed to the fp16 scheduler (table cast to fp16 first).
    """
    n = c.shape[0]
    noise = noise.expand(n, -1, -1, -1)
    xe = x.expand(n, -1, -1, -1)
    te = timesteps.expand(n)
    if autocast and latent_dtype == torch.float16:
        noise = noise.half()
        noisy = add_noise(xe.half(), noise, te, acp).float()
    else:
        noisy = add_noise(xe.float(), noise.float(), te, acp)        # fp32; unet_forward rounds it when autocast
    pred = unet_forward(sd, noisy, te, c, cfg, autocast)
    return F.mse_loss(pred.float(), noise.float(), reduction="none")


def draw_noise_and_timesteps(shape, N, t_min, t_max, seed=42, num_train_timesteps=1000,
                             dtype=torch.float32):
    """`D.noising` ×N after `torch.manual_seed(seed)` (compute.py:115-124,139-141), on the CPU
    generator: interleaved randn_like / randint draws.  Device (Philox) draws are launch-geometry
    dependent and not portable, so parity tests inject these explicitly."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    lo, hi = int(t_min * num_train_timesteps), int(t_max * num_train_timesteps)
    noises, ts = [], []
    for _ in range(N):
        noises.append(torch.randn(shape, generator=g, dtype=torch.float32).to(dtype))
        ts.append(torch.randint(lo, hi, (1,), generator=g).long())
    return torch.cat(noises, 0), torch.cat(ts, 0)


def compute_losses(sd, x, cond_embeds, noises, timesteps, B=10, cfg: RefConfig = SD15_REF,
                   autocast: bool = True, latent_dtype: torch.dtype = torch.float32):
    """`D.compute_losses` (compute.py:134-160) from the latent on (VAE is outside the path).

    x [1,4,h,w]; cond_embeds [n_cond,77,768] (index 0 = c, 1 = null, compute.py:187-188);
    noises [N,4,h,w], timesteps [N].  Returns [N, n_cond, 4, h, w] float16.
    """
    grids = []
    n_cond = cond_embeds.shape[0]
    for i in range(0, noises.shape[0], B):
        nb, tb = noises[i:i + B], timesteps[i:i + B]
        bs = nb.shape[0]
        n_batch = torch.cat([nb] * n_cond, 0)                       # cond-major tiling (:150-151)
        t_batch = torch.cat([tb] * n_cond, 0)
        c = torch.cat([cond_embeds[k].unsqueeze(0).expand(bs, -1, -1) for k in range(n_cond)], 0)
        loss = compute_loss(sd, x, n_batch, t_batch, c, cfg, autocast, latent_dtype=latent_dtype)
        grids.append(torch.stack(torch.split(loss, [bs] * n_cond, dim=0), dim=1))   # (:155)
    return torch.cat(grids, 0).to(torch.float16)


def typicality_map(grid: torch.Tensor) -> torch.Tensor:
    """Per-latent-pixel E_N[L_null - L_c] after the latent-channel mean: the quantity
    `Typicallity.compute` (applications/xray/compute.py:210-218) forms before/without the
    bilinear resize.  grid [N,2,4,h,w] -> [h,w] fp32."""
    dm = grid.float().mean(dim=2)
    return (dm[:, -1] - dm[:, 0]).mean(0)


def load_typicality(grid: torch.Tensor, image_size, kx: int, ky: int) -> torch.Tensor:
    """`Cluster.load_typicality` (cluster.py:125-137) + `pool` (utils.py:74-80), in the reference's
    own order of operations: mean over C, bilinear resize to (H, W), AvgPool2d((kx, ky), stride 1) per
    condition, -(pool(c) - pool(null)), mean over N.  grid [N,2,4,h,w] -> [H-kx+1, W-ky+1]."""
    dm = grid.float().mean(dim=2)
    dm = F.interpolate(dm, tuple(image_size), mode="bilinear")

    def pool(x):
        if kx != 1 and ky != 1:
            return torch.nn.AvgPool2d((kx, ky), stride=(1, 1), padding=0)(x)
        return x
    d = pool(dm[:, 0].unsqueeze(1)) - pool(dm[:, 1].unsqueeze(1))
    return -d.squeeze(1).mean(dim=0)


def normalize_map(dm, mode: str = "signed"):
    """The consumers' normalisations of a map (numpy fp32, restated): "signed" = `normalize(dm)` of cluster.py:32-47 with
    positive_only=False (negatives / |min|, positives / max, (dm + 1) / 2), "positive" = positive_only=True (cluster.py:39-42,
    utils.py:16-19), "split" = positive_only='split' (cluster.py:34-36), "maxabs" = `dm / np.max(np.abs(dm))` (utils.py:14-20,
    `d_compute` utils.py:130).  Pinned to the reference's own functions by tests/golden/consumers_ref.npz."""
    import numpy as np
    dm = np.array(dm, dtype=np.float32, copy=True)
    if mode == "split":
        dm = dm / np.abs(np.max(dm))
        return np.clip(dm, 0, 1), -np.clip(dm, -1, 0)
    if mode == "positive":
        dm = np.maximum(dm, 0)
        return dm / np.max(dm)
    if mode == "maxabs":
        return dm / np.max(np.abs(dm))
    assert mode == "signed", mode
    lo = np.abs(np.min(dm))
    neg = dm < 0
    dm[neg] = dm[neg] / lo
    pos = dm > 0
    dm[pos] = dm[pos] / np.max(dm)
    return (dm + np.float32(1)) / np.float32(2.0)


def load_typicality_norm(grid: torch.Tensor, image_size):
    """`Cluster.load_typicality_norm` (cluster.py:112-123): mean over C, bilinear to (H, W), `(dm[:, 1] - dm[:, 0]).mean(0)`,
    `normalize`.  grid [N,2,4,h,w] -> numpy [H, W] fp32 in [0, 1]."""
    dm = F.interpolate(grid.float().mean(dim=2), tuple(image_size), mode="bilinear")
    return normalize_map((dm[:, 1] - dm[:, 0]).mean(dim=0).numpy(), "signed")


def d_compute(grid: torch.Tensor, h: int, w: int, x_start: int, y_start: int, x_end: int, y_end: int):
    """`d_compute` (utils.py:122-134): the same per-pixel map at (h, w), divided by its max |.|, cropped."""
    dm = F.interpolate(grid.float().mean(dim=2), (h, w), mode="bilinear")
    dm = (dm[:, 1] - dm[:, 0]).mean(dim=0).numpy()
    return normalize_map(dm, "maxabs")[x_start:x_end, y_start:y_end]


def typicality_scalar(grid: torch.Tensor) -> torch.Tensor:
    """T(x|c) = mean over pixels of `typicality_map` (intent of cluster.py:517-531)."""
    return typicality_map(grid).mean()


# --------------------------------------------------------------------------------------------
# DIFT surface — diffmining/typicality/dift.py:173-232
# --------------------------------------------------------------------------------------------
def dift_features(sd, latents_noisy, t, prompt_embeds, up_ft_index=1, cfg: RefConfig = SD15_REF,
                  autocast: bool = False):
    """`MyUNet2DConditionModel.forward(latents_noisy, t, [up_ft_index], prompt_embeds)` then the
    ensemble mean of `SDFeaturizer.forward` (dift.py:229-231).  Reference runs this in fp32."""
    tt = torch.as_tensor(t, dtype=torch.long)
    out = unet_forward(sd, latents_noisy, tt, prompt_embeds, cfg, autocast, up_ft_indices=[up_ft_index])
    ft = out["up_ft"][up_ft_index]
    return ft, ft.mean(0, keepdim=True)


# --------------------------------------------------------------------------------------------
# DIFT patch descriptor — `Cluster.compute_embeddings`, cluster.py:291-299
# --------------------------------------------------------------------------------------------
def dift_patch_embedding(feat, box_px, image_hw):
    """feat [C,h,w] (the ensemble-mean DIFT map, numpy), box_px = (x_start, y_start, x_end, y_end) in
    image pixels with x = rows, y = columns (cluster.py:258-262), image_hw = (image.height, image.width).
    `emb[:, int(x0*H):int(x1*H), int(y0*W):int(y1*W)].mean(axis=(1,2)); emb / ||emb||`."""
    import numpy as np
    _, h, w = feat.shape
    H = h / image_hw[0]
    W = w / image_hw[1]
    x0, y0, x1, y1 = box_px
    emb = feat[:, int(x0 * H):int(x1 * H), int(y0 * W):int(y1 * W)]
    emb = emb.mean(axis=(1, 2))
    return emb / np.linalg.norm(emb)
