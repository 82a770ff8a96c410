"""Host-side mirror of the reference's DIFT featuriser (diffmining/typicality/dift.py:173-232) over
the HIP engine.

`SDFeaturizer.forward(img_tensor, prompt, t=261, up_ft_index=1, ensemble_size=8)` keeps the reference's
argument meaning and its output `[1, C, H/16, W/16]` (for up_ft_index=1):
  * `img_tensor` with 3 channels is an image in [-1, 1] like dift.py:214-232 (needs VAE weights on the engine: one
    encoder pass, `ensemble_size` posterior samples — the reference encodes the same image `ensemble_size` times);
    with 4 channels it is the scaled VAE latent (callers that hold latents);
  * `prompt` is a string like the reference's (`pipe.encode_prompt`, dift.py:222-226: tokenizer with
    padding="max_length", CLIP text tower, last hidden state) when the featurizer was built with a tokenizer and the
    engine holds CLIP text weights, or the CLIP hidden states `[1,77,768]` themselves.
`patch_embeddings` is the DIFT branch of `Cluster.compute_embeddings` (cluster.py:288-299) with a
per-image cache of the feature map.

Arithmetic: the reference's featuriser is fp32 end to end (dift.py:197-199: no torch_dtype; :191: no autocast).  Built over a
`UNetEngineF32` the U-Net runs in that arithmetic (`self.dtype == torch.float32`, matches the CPU oracle in fp32 mode to ~1e-6); built over
the fp16 `UNetEngine` it is the fast reduced-precision mode (descriptor cosine >= 1 - 5e-7), which also
carries the VAE encoder and the CLIP text tower for image / string inputs.  `SDFeaturizer(f32_net, aux=fp16_engine)` combines
them: fp32 U-Net, the fp16 engine only for `vae_encode` / `clip_encode` / `patch_embed`; an fp32 net that holds its own VAE / CLIP
weights (`load_vae_state_dict`, `load_clip_state_dict`) runs those stages in fp32 too, as the reference does.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Sequence, Tuple

import torch

from .engine import UNetEngine, UNetEngineF32


def scheduler_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def feature_boxes(boxes_px: Sequence[Tuple[int, int, int, int]], image_hw: Tuple[int, int],
                  feat_hw: Tuple[int, int]) -> torch.Tensor:
    """Pixel boxes (x_start, y_start, x_end, y_end; x = rows, y = columns, cluster.py:258-262) -> feature
    cells (r0, r1, c0, c1) with the reference's arithmetic `int(x * (h / image.height))` (cluster.py:292-296)."""
    H = feat_hw[0] / image_hw[0]
    W = feat_hw[1] / image_hw[1]
    return torch.tensor([[int(x0 * H), int(x1 * H), int(y0 * W), int(y1 * W)] for (x0, y0, x1, y1) in boxes_px],
                        dtype=torch.int32).reshape(-1, 4)


class SDFeaturizer:
    def __init__(self, engine, cache_size: int = 64, tokenizer=None, aux: Optional[UNetEngine] = None):
        """`engine`: a `UNetEngineF32` (the reference's fp32 arithmetic) or a `UNetEngine` (fp16, faster); `aux`: the fp16
        engine that holds the VAE encoder / CLIP text tower / patch kernel when `engine` is the fp32 net;
        `tokenizer`: e.g. transformers' `CLIPTokenizer` (dift.py:203); only needed for string prompts."""
        self.engine = engine
        self.dtype = torch.float32 if isinstance(engine, UNetEngineF32) else torch.float16
        self.aux = aux if aux is not None else (None if isinstance(engine, UNetEngineF32) else engine)
        self.tokenizer = tokenizer
        self._prompt_cache: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.device = engine.device
        self.acp = scheduler_alphas_cumprod().to(self.device)
        self._cache: "OrderedDict[object, torch.Tensor]" = OrderedDict()   # key -> ensemble-mean map on the GPU
        self.cache_size = cache_size
        self._registered = None             # (engine.prompt_generation, embeds fp16 [1,77,768]) of the last set_prompts made here
        self.prompt_registrations = 0

    @staticmethod
    def _version_of(t):
        """`t._version`, or None for tensors without a version counter (created under `torch.inference_mode()`):
        the identity fast path is skipped for them and the value comparison decides."""
        try:
            return None if t.is_inference() else t._version
        except RuntimeError:
            return None

    def _register_prompt(self, prompt_embeds):
        """`set_prompts` (the 16 blocks' cross-attention K/V projections of the prompt) only when the engine does not hold
        this prompt already: the reference featurises up to five patches per image under ONE category prompt (cluster.py:224-
        226,291), so the projections are the same from call to call.  Anyone else's `set_prompts` on the shared engine bumps
        `prompt_generation` and forces a re-registration."""
        eng = self.engine
        r = self._registered
        if r is not None and r[0] == eng.prompt_generation:
            if r[1] is prompt_embeds and r[2] is not None and r[2] == self._version_of(prompt_embeds):   # the cached tensor of a string prompt
                return
            pe16 = prompt_embeds.reshape(1, 77, -1).to(self.device, self.dtype)
            if r[3].shape == pe16.shape and torch.equal(r[3], pe16):             # a caller's fresh tensor with the same values
                self._registered = (r[0], prompt_embeds, self._version_of(prompt_embeds), r[3])
                return
        pe = prompt_embeds.reshape(1, 77, -1)
        eng.set_prompts(pe)
        self.prompt_registrations += 1
        self._registered = (eng.prompt_generation, prompt_embeds, self._version_of(prompt_embeds), pe.to(self.device, self.dtype).clone())

    def add_noise(self, latents, noise, t):
        """DDIMScheduler.add_noise in the latents' dtype (dift.py:190; fp32 in the reference)."""
        a = self.acp.to(latents.dtype)[t]
        return (a ** 0.5) * latents + ((1 - a) ** 0.5) * noise

    @torch.no_grad()
    def encode_prompt(self, prompt):
        """`pipe.encode_prompt(prompt, do_classifier_free_guidance=False)[0]` (dift.py:222-226) -> [1,77,768] fp32 on the
        GPU: tokenizer (padding="max_length", truncation) on the host, CLIP text tower on the engine; one entry per
        distinct string is kept (the reference re-encodes the category prompt for every patch, cluster.py:224-226)."""
        if torch.is_tensor(prompt):
            return prompt.reshape(1, 77, -1)
        if not isinstance(prompt, str):
            raise TypeError(f"prompt must be a str or a [1,77,768] tensor, got {type(prompt).__name__}")
        hit = self._prompt_cache.get(prompt)
        if hit is not None:
            self._prompt_cache.move_to_end(prompt)
            return hit
        # the fp32 net's own text tower when it holds the weights (r06: `pipe.encode_prompt` of the reference's featuriser is fp32,
        # dift.py:197-199, 222-226 — 2e-6 from transformers' fp32 output), else the fp16 engine's (1.1e-3 from it, DESIGN.md section 2:
        # the one reduced-precision stage of an otherwise fp32 featuriser); hidden states passed as a tensor are used as they are
        tower = self.engine if (isinstance(self.engine, UNetEngineF32) and getattr(self.engine, "_clip_ready", False)) else self.aux
        if self.tokenizer is None or tower is None:
            raise ValueError("a string prompt needs SDFeaturizer(engine, tokenizer=...) and CLIP text weights (engine.load_clip_state_dict on "
                             "the fp32 net, or on the fp16 engine passed as `aux=`); pass the [1,77,768] hidden states otherwise")
        tok = self.tokenizer
        ids = tok([prompt], max_length=tok.model_max_length, padding="max_length", truncation=True, return_tensors="pt").input_ids
        emb = tower.clip_encode(ids)
        self._prompt_cache[prompt] = emb
        while len(self._prompt_cache) > 256:
            self._prompt_cache.popitem(last=False)
        return emb

    @torch.no_grad()
    def forward(self, img_tensor, prompt, t: int = 261, up_ft_index: int = 1, ensemble_size: int = 8,
                noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None, vae_noise=None):
        """`SDFeaturizer.forward` (dift.py:214-232).  img_tensor: image [1,3,H,W] / [3,H,W] in [-1,1] (VAE-encoded here),
        or latents [1,4,h,w] (repeated `ensemble_size` times, dift.py:220) / [ensemble_size,4,h,w] (independent
        posterior draws, dift.py:187); prompt: str or hidden states [1,77,768].
        Returns the ensemble-mean feature [1, C, h', w'] fp32 (dift.py:231)."""
        prompt_embeds = self.encode_prompt(prompt)
        if img_tensor.shape[-3] == 3:
            return self.forward_image(img_tensor, prompt_embeds, t, up_ft_index, ensemble_size, vae_noise=vae_noise, noise=noise,
                                      generator=generator)
        latents = img_tensor
        lat = latents.to(self.device, torch.float32)
        if lat.shape[0] == 1:
            lat = lat.repeat(ensemble_size, 1, 1, 1)
        assert lat.shape[0] == ensemble_size
        if noise is None:
            noise = torch.randn(lat.shape, generator=generator, dtype=torch.float32,
                                device=generator.device if generator is not None else "cpu")
        noisy = self.add_noise(lat, noise.to(self.device, torch.float32), int(t))
        self._register_prompt(prompt_embeds)
        slots = torch.zeros(ensemble_size, dtype=torch.int32, device=self.device)
        _, mean = self.engine.dift(noisy.to(self.dtype), torch.tensor(int(t)), slots, up_ft_index, ensemble_size)
        return mean

    __call__ = forward

    @torch.no_grad()
    def forward_image(self, img_tensor, prompt_embeds, t: int = 261, up_ft_index: int = 1, ensemble_size: int = 8,
                      vae_noise=None, noise=None, generator: Optional[torch.Generator] = None):
        """`SDFeaturizer.forward(img_tensor, ...)` from pixels (dift.py:214-232): the reference encodes the
        image `ensemble_size` times and takes one posterior sample of each (dift.py:220,187); here the VAE
        encoder runs once and `ensemble_size` posterior samples are drawn from its moments (same distribution).
        img_tensor [3,H,W] or [1,3,H,W] in [-1,1] (`dift_pre`, dift.py:19-21).  Needs VAE weights."""
        prompt_embeds = self.encode_prompt(prompt_embeds)
        img = img_tensor if img_tensor.dim() == 4 else img_tensor[None]
        _, _, H, W = img.shape
        if vae_noise is None:
            vae_noise = torch.randn(ensemble_size, 4, H // 8, W // 8, generator=generator, dtype=torch.float32)
        if self.dtype == torch.float32 and getattr(self.engine, "_vae_ready", False):
            # the reference's arithmetic end to end: fp32 VAE encoder (dift.py:187 on the fp32 pipeline of dift.py:197-199)
            lat = self.engine.vae_encode(img, vae_noise, draws_per_image=ensemble_size)
        elif self.aux is not None:
            lat = self.aux.vae_encode(img, vae_noise.to(torch.float16), out_dtype=torch.float32, draws_per_image=ensemble_size)
        else:
            raise ValueError("an image input needs VAE weights: f32_net.load_vae_state_dict(...) or SDFeaturizer(f32_net, aux=fp16_engine)")
        return self.forward(lat, prompt_embeds, t, up_ft_index, ensemble_size, noise=noise, generator=generator)

    # -- Cluster.compute_embeddings' DIFT branch (cluster.py:288-299) ------------------------------
    @torch.no_grad()
    def patch_embeddings(self, feat_or_key, boxes_px, image_hw, compute=None):
        """L2-normalised window-mean descriptors [P, C] (fp32, GPU) of all patches of ONE image in a single
        launch.  The reference re-runs the full DIFT forward for every patch row of the dataframe
        (cluster.py:291: up to 5 per image); here the ensemble-mean map is computed once per image and kept
        in a small LRU cache: `feat_or_key` is either the map itself ([1,C,h,w] / [C,h,w]) or a hashable
        cache key (e.g. (image path, prompt, t)) with `compute()` -> map called on a miss."""
        if torch.is_tensor(feat_or_key):
            feat = feat_or_key
        else:
            feat = self._cache.get(feat_or_key)
            if feat is None:
                assert compute is not None, "cache miss and no compute() given"
                feat = compute().to(self.device, torch.float32)
                self._cache[feat_or_key] = feat
                while len(self._cache) > self.cache_size:
                    self._cache.popitem(last=False)
            else:
                self._cache.move_to_end(feat_or_key)
        fh, fw = feat.shape[-2], feat.shape[-1]
        if self.aux is None:
            raise ValueError("patch_embeddings needs the fp16 engine's patch kernel: SDFeaturizer(f32_net, aux=fp16_engine)")
        return self.aux.patch_embed(feat, feature_boxes(boxes_px, image_hw, (fh, fw)))
