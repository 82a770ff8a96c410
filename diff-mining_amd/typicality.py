"""Host-side mirror of the reference's scoring surface over the HIP engine.

    reference                                         here
    ------------------------------------------------  -----------------------------------------
    CategoryFeatures         compute.py:27-54         CategoryFeatures (optional: needs CLIP text weights)
    SD.encode_vae            compute.py:91-93         TypicalityScorer.encode_vae (optional: needs VAE weights)
    D.load_image             compute.py:126-132       TypicalityScorer.load_image
    SD.compute_loss          compute.py:95-102        TypicalityScorer.compute_loss
    D.noising                compute.py:115-124       TypicalityScorer.noising / draw
    D.compute_losses         compute.py:134-160       TypicalityScorer.compute_losses
    compute_submission loop  compute.py:284-290       TypicalityScorer.compute_submission -> compute_losses_batch (n images, each under
                                                      its own category, one engine call)
    D.get_path / np.save     compute.py:162-163,192   TypicalityScorer.save_grid (same .npy layout)
    D.rescale                compute.py:165-180       TypicalityScorer.rescale
    D.compute / __call__ / exists  compute.py:182-202 TypicalityScorer.compute / __call__ / exists
    Typicallity.compute      xray/compute.py:210-218  TypicalityScorer.heatmap / typicality_scalar
    Cluster.load_typicality(_norm) cluster.py:112-137 TypicalityScorer.load_typicality / load_typicality_norm
    d_compute                utils.py:122-134         TypicalityScorer.d_compute
    unet(sample, t, c).sample compute.py:100          UNetCallable (assignable over `pipe.unet`)

Same names, argument meaning and output layout ([N, n_cond, 4, h, w] float16, cond 0 = c,
1 = null); differences are stated where they exist:
  * `compute_losses` takes the latent `x`; `compute_losses_from_image` runs the VAE encode of
    compute.py:137 on the engine too (the posterior draw is injected: the reference's is unseeded);
  * all N draws of an image are scored in as few U-Net batches as the engine's workspace allows
    instead of B-sized chunks with a D2H copy each (compute.py:145-156) — `B` is accepted and
    ignored for the result (it never changes the math, only the chunking);
  * (eps, t) are drawn on the CPU generator by default so the values are reproducible across
    devices; the reference's device-Philox draws are launch-geometry dependent.
  * dtype flow (`latent_dtype`, default torch.float32 = the reference's): under `@torch.autocast` the VAE
    posterior's `exp` is promoted, so `encode_vae` returns an fp32 latent (compute.py:91-93), `randn_like(x)`
    is fp32 (:116), `scheduler.add_noise` runs in fp32 on the fp32 table (:99) and `mse_loss` sees the fp32
    eps (:101); only the U-Net input is rounded to fp16.  `latent_dtype=torch.float16` selects an all-fp16
    add_noise (table cast to fp16 first) for callers that hold fp16 latents.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional, Sequence

import numpy as np
import torch

from .engine import UNetEngine


class CategoryFeatures:
    """`CategoryFeatures` of compute.py:27-54 over the engine's CLIP text tower: one prompt per category
    (templates of compute.py:41-48, '' = the null prompt; which = "xray": the X-ray application's `Embed`,
    applications/xray/compute.py:40-60, whose null prompt is the non-empty "Chest X-Ray"), tokenised on the host by `tokenizer` (e.g.
    transformers' `CLIPTokenizer`, called exactly like compute.py:36-37), encoded on the GPU."""

    def __init__(self, engine: UNetEngine, tokenizer, which: str):
        self.engine, self.tokenizer, self.which = engine, tokenizer, which

    @staticmethod
    def prompts(which: str, categories):
        if which == "faces":
            return [(f"Portrait at the {c}'s." if len(c) else "Portrait.") for c in categories]
        if which == "cars":
            return [(f"A car at the {c}'s." if len(c) else "A car.") for c in categories]
        if which == "places":
            return [("Image of " + c.replace("_", " ") + "." if len(c) else "") for c in categories]
        if which == "xray":      # `Embed.embed_diseases` (applications/xray/compute.py:54-57): the null prompt is NOT empty
            return [(f"Chest X-Ray with {c}." if len(c) else "Chest X-Ray") for c in categories]
        return [(f"{c}" if len(c) else "") for c in categories]

    def tokenize(self, prompts):
        return self.tokenizer(prompts, max_length=self.tokenizer.model_max_length, padding="max_length", truncation=True,
                              return_tensors="pt").input_ids

    @torch.no_grad()
    def embed(self, categories):
        return self.engine.clip_encode(self.tokenize(self.prompts(self.which, categories)))     # [n,77,768] fp32

    def __getitem__(self, x):
        return self.embed(x)


class _Sample:
    """`unet(...).sample` result object."""

    def __init__(self, sample):
        self.sample = sample


class UNetCallable:
    """Drop-in for `pipe.unet` on the scoring path: `unet(sample, timestep, encoder_hidden_states).sample`.

    Distinct prompts are detected by row equality (the reference tiles n_cond prompts over 2B rows,
    compute.py:152); their cross-attention K/V are computed once."""

    def __init__(self, engine: UNetEngine):
        self.engine = engine
        self._ctx_key = None                # [P, 77*768] fp16: the distinct prompts registered with the engine
        self._ctx_generation = -1
        self._key_hash = None               # [P] int64 row hashes of _ctx_key
        self._hash_w = None
        self._last = None                   # (source tensor, its _version, slots): the identical-object fast path
        self.stats = {"identity_hits": 0, "key_hits": 0, "unique_calls": 0}

    def _row_hash(self, flat):
        """64-bit hash per row of an fp16 [n, L] matrix (L even): the rows as int32 words against fixed odd weights."""
        words = flat.view(torch.int32).to(torch.int64)
        if self._hash_w is None or self._hash_w.shape[0] != words.shape[1] or self._hash_w.device != flat.device:
            g = torch.Generator().manual_seed(0x5D1F)
            self._hash_w = (torch.randint(-(1 << 40), 1 << 40, (words.shape[1],), generator=g, dtype=torch.int64) | 1).to(flat.device)
        return (words * self._hash_w).sum(1)

    @staticmethod
    def _version_of(c):
        """`c._version`, or None where tensors carry no version counter (created under `torch.inference_mode()`): the identity
        fast path is then skipped and the row-hash path decides."""
        try:
            return None if c.is_inference() else c._version
        except RuntimeError:
            return None

    def _unique_rows(self, flat):
        """Distinct rows of flat [n, L] and the row -> distinct-row map.  `torch.unique(dim=0)` sorts the rows lexicographically — 104 ms
        on the GPU for 16 prompts of 77 x 768 (r05 profile: one rocprim block sort, the longest kernel of a bench run) — so the rows
        are grouped by their 64-bit hash (a 1-D unique of n values) and the grouping is confirmed by ONE exact comparison; a hash
        collision (never seen) falls back to the sort."""
        n = flat.shape[0]
        if flat.dtype == torch.float16 and flat.shape[1] % 2 == 0 and n > 0:
            uh, inv = torch.unique(self._row_hash(flat), return_inverse=True)
            first = torch.full((uh.numel(),), n, dtype=torch.long, device=flat.device).scatter_reduce_(
                0, inv, torch.arange(n, device=flat.device), reduce="amin")
            order = torch.argsort(first)                      # slots in order of first appearance
            rank = torch.empty_like(order)
            rank[order] = torch.arange(order.numel(), device=flat.device)
            uniq, inv = flat[first[order]], rank[inv]
            if bool((flat == uniq[inv]).all()):
                return uniq, inv
        return torch.unique(flat, dim=0, return_inverse=True)

    def _slots_for(self, c, ident=None):
        """Prompt slot of every row of c [n, 77, 768] (fp16, on the device; or a callable that builds it — only called when the
        identity path misses).  `ident`: the caller's own tensor object when `c` is a derived view / cast of it (default: c).
        Three paths, cheapest first:
          1. the same tensor object, unmodified, as the previous call (a caller that keeps its `c`): nothing runs;
          2. every row equals one of the prompts already registered (the reference builds a fresh `torch.cat` of the same
             n_cond embeddings for every chunk of every image, compute.py:152): a row hash finds the candidate slot, ONE exact
             comparison against the registered rows confirms it — no sort, one scalar read-back;
          3. otherwise `torch.unique(dim=0)` (sort + sync) and a new `set_prompts`."""
        eng = self.engine
        # the engine's K/V cache is shared: anyone else's set_prompts (SDFeaturizer, compute_losses, a direct
        # call) bumps `prompt_generation`, which invalidates every cache here
        valid = self._ctx_key is not None and self._ctx_generation == eng.prompt_generation
        ident = c if ident is None else ident
        if valid and self._last is not None and self._last[0]() is ident and self._last[1] == self._version_of(ident):
            self.stats["identity_hits"] += 1
            return self._last[2]
        if callable(c):
            c = c()
        flat = c.reshape(c.shape[0], -1)
        inv = None
        if valid and flat.shape[1] == self._ctx_key.shape[1] and flat.shape[1] % 2 == 0 and flat.dtype == torch.float16:
            cand = (self._row_hash(flat)[:, None] == self._key_hash[None, :]).to(torch.uint8).argmax(1)
            if bool((flat == self._ctx_key[cand]).all()):
                inv = cand.to(torch.int32)
                self.stats["key_hits"] += 1
        if inv is None:
            self.stats["unique_calls"] += 1
            uniq, inv = self._unique_rows(flat)
            eng.set_prompts(uniq.reshape(uniq.shape[0], c.shape[1], c.shape[2]))
            self._ctx_key = uniq
            self._key_hash = self._row_hash(uniq) if uniq.shape[1] % 2 == 0 and uniq.dtype == torch.float16 else None
            self._ctx_generation = eng.prompt_generation
            inv = inv.to(torch.int32)
            if self._key_hash is None:
                self._ctx_key = None
        v = self._version_of(ident)
        self._last = (weakref.ref(ident), v, inv) if v is not None else None   # a weak reference: the cache must not keep the caller's tensor alive
        return inv

    def __call__(self, sample, timestep, encoder_hidden_states, **_):
        eng = self.engine
        c = encoder_hidden_states.to(eng.device, torch.float16)
        slots = self._slots_for(c)
        t = torch.as_tensor(timestep, device=eng.device).reshape(-1)
        if t.numel() == 1:
            t = t.expand(sample.shape[0])
        return _Sample(eng.unet(sample, t, slots))


class TypicalityScorer:
    """`SD` + `D` of diffmining/typicality/compute.py:56-160 on the MI355X engine."""

    def __init__(self, engine: UNetEngine, seed: int = 42, N: int = 100, t_min: float = 0.0, t_max: float = 1.0,
                 num_train_timesteps: int = 1000, generator_device: str = "cpu", latent_dtype=torch.float32,
                 typicality_path: Optional[str] = None, which: Optional[str] = None, country_embeds=None):
        """`D(sd, typicality_path, which, seed, N, t_min, t_max)` (compute.py:105-113).  `country_embeds`: the
        `SD.country_embeds` dict {category or "": [77,768]} (compute.py:76-80) that `compute(country, path)` reads."""
        self.engine = engine
        self.device = engine.device
        self.seed, self.N, self.t_min, self.t_max = seed, N, t_min, t_max
        self.num_train_timesteps = num_train_timesteps
        self.generator_device = generator_device
        assert latent_dtype in (torch.float32, torch.float16)
        self.latent_dtype = latent_dtype
        self.typicality_path, self.which, self.country_embeds = typicality_path, which, country_embeds
        self.unet = UNetCallable(engine)
        self.last_loss32 = None

    # -- D.load_image / SD.encode_vae (compute.py:126-132, 91-93) --------------------------------
    @staticmethod
    def load_image(x) -> torch.Tensor:
        """PIL image or uint8 HWC array -> [1,3,H,W] float in [-1,1] (`to_tensor(x) * 2 - 1`)."""
        a = np.asarray(x.convert("RGB") if hasattr(x, "convert") else x)
        assert a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8, (a.shape, a.dtype)
        return (torch.from_numpy(a.copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1).unsqueeze(0)

    @torch.no_grad()
    def encode_vae(self, x, noise=None, generator: Optional[torch.Generator] = None, scaling_factor: float = 0.18215):
        """`vae.encode(x).latent_dist.sample() * scaling_factor` -> [B,4,H/8,W/8] latents on the GPU, in
        `self.latent_dtype` (fp32 like the reference: fp16 mean + fp32 std * fp16 draw, times the factor in fp32).
        The posterior draw: `noise` if given, else N(0,1) from `generator` (CPU) — the reference draws it
        unseeded on the device before `manual_seed(seed)` (compute.py:137-139), in the moments' fp16."""
        B, _, H, W = x.shape
        if noise is None:
            noise = torch.randn(B, 4, H // 8, W // 8, generator=generator, dtype=torch.float32).to(torch.float16)
        return self.engine.vae_encode(x, noise, scaling_factor, out_dtype=self.latent_dtype)

    @torch.no_grad()
    def compute_losses_from_image(self, img, country_embeds, B: int = 10, vae_noise=None, to_host: bool = True):
        """`D.compute_losses(img, country_embeds)` from pixels: VAE encode, then the N x n_cond scoring grid."""
        x = self.encode_vae(img if torch.is_tensor(img) else self.load_image(img), vae_noise)
        return self.compute_losses(x, country_embeds, B, to_host=to_host)

    # -- SD.compute_loss (compute.py:95-102) -----------------------------------------------------
    @torch.no_grad()
    def compute_loss(self, x, noise, timesteps, c):
        """x [1 or 2B,4,h,w]; noise [2B,4,h,w]; timesteps [2B]; c [2B,77,768] -> loss [2B,4,h,w] fp32."""
        n = c.shape[0]
        noise = noise.expand(n, -1, -1, -1)
        timesteps = timesteps.expand(n)
        slots = self.unet._slots_for(c.to(self.device, torch.float16))
        return self.engine.score(x, noise, timesteps, slots, latent_dtype=self.latent_dtype)

    # -- D.noising (compute.py:115-124) ----------------------------------------------------------
    def draw(self, shape, N: Optional[int] = None):
        """N interleaved (randn_like, randint) draws after manual_seed(seed) (compute.py:139-141); eps in
        `self.latent_dtype` (`randn_like(x)` of the fp32 latent is fp32; the fp16 flow rounds the same draws)."""
        N = self.N if N is None else N
        g = torch.Generator(device=self.generator_device)
        g.manual_seed(self.seed)
        lo, hi = int(self.t_min * self.num_train_timesteps), int(self.t_max * self.num_train_timesteps)
        noises, ts = [], []
        for _ in range(N):
            noises.append(torch.randn(tuple(shape), generator=g, dtype=torch.float32,
                                      device=self.generator_device).to(self.latent_dtype))
            ts.append(torch.randint(lo, hi, (1,), generator=g, device=self.generator_device).long())
        return torch.cat(noises, 0), torch.cat(ts, 0)

    # -- D.compute_losses (compute.py:134-160) ---------------------------------------------------
    @torch.no_grad()
    def compute_losses(self, x, country_embeds, B: int = 10, noises=None, timesteps=None, to_host: bool = True):
        """x [1,4,h,w] latent; country_embeds [n_cond,77,768] (0 = c, 1 = null).
        Returns [N, n_cond, 4, h, w] float16 (on the host like the reference, or on the GPU)."""
        eng = self.engine
        if noises is None or timesteps is None:
            noises, timesteps = self.draw(x.shape)
        N = noises.shape[0]
        n_cond = country_embeds.shape[0]
        eng.set_prompts(country_embeds)
        # sample row k*N + i = draw i under condition k  -> view as [n_cond, N] then transpose
        if n_cond >= 2:
            loss = eng.score_conds(x, noises, timesteps, n_cond, latent_dtype=self.latent_dtype)   # cond-major rows
        else:
            slots = torch.zeros(N, dtype=torch.int32)
            loss = eng.score(x, noises, timesteps, slots, latent_dtype=self.latent_dtype)
        grid = loss.view(n_cond, N, *loss.shape[1:]).transpose(0, 1).to(torch.float16)   # compute.py:155,160
        return grid.cpu() if to_host else grid.contiguous()

    # -- D.compute_losses over a slice of the work list (compute.py:284-290, 182-192) ------------
    @torch.no_grad()
    def compute_losses_batch(self, xs, country_embeds, B: int = 10, noises=None, timesteps=None, to_host: bool = True):
        """`D.compute_losses` for n images of one latent size in ONE engine call — what `compute_submission`'s loop over the
        `path,category` lines of a work list (compute.py:284-290) does image by image, each image under ITS OWN category and the
        shared null prompt (`D.compute`, compute.py:182-192).
          xs [n,4,h,w] latents;  country_embeds [n_cond,77,768] (one prompt set for every image) or [n,n_cond,77,768] (per image;
          repeated prompts — the null prompt, images of one category — are registered once);
          noises / timesteps: None = D's own draws: `manual_seed(seed)` precedes every image's N draws (compute.py:139-141), so
          images of one size see the SAME N (eps, t); or [N,4,h,w] / [N] shared, or [n,N,4,h,w] / [n,N] per image.
        Returns [n, N, n_cond, 4, h, w] float16 — image j's slice is bit-equal to `compute_losses(xs[j:j+1], embeds_j)`: the engine
        guarantees that a sample's bits do not depend on the batch it rides in; the prompt-independent head of the U-Net still
        runs once per (image, draw)."""
        eng = self.engine
        n = xs.shape[0]
        if noises is None or timesteps is None:
            noises, timesteps = self.draw(xs[0:1].shape)
        per_image_draws = noises.dim() == 5
        N = noises.shape[1] if per_image_draws else noises.shape[0]
        if per_image_draws:
            assert noises.shape[0] == n and tuple(timesteps.shape) == (n, N), (noises.shape, timesteps.shape)
            eps_all, t_all = noises.reshape((n * N,) + tuple(noises.shape[2:])), timesteps.reshape(n * N)
        else:
            eps_all, t_all = noises.repeat(n, 1, 1, 1), timesteps.repeat(n)          # image-major: row j*N + i = draw i of image j
        x_index = torch.arange(n, dtype=torch.int32, device=self.device).repeat_interleave(N)       # built on the device: no H2D copy in the loop
        per_image_prompts = country_embeds.dim() == 4
        n_cond = country_embeds.shape[1] if per_image_prompts else country_embeds.shape[0]
        table = None
        if per_image_prompts:
            assert country_embeds.shape[0] == n, (country_embeds.shape, n)
            def flat_c():
                return country_embeds.reshape((n * n_cond,) + tuple(country_embeds.shape[2:])).to(self.device, torch.float16)
            # registers the distinct prompts; a caller that passes the same tensor again (a work list walked in batches of one
            # category set, bench.py) takes the identity path: no hashing, no read-back
            inv = self.unet._slots_for(flat_c, ident=country_embeds).reshape(n, n_cond)
            table = inv.t().repeat_interleave(N, dim=1).contiguous()                 # [n_cond, n*N]: slot of draw (j, i) under condition k
        else:
            eng.set_prompts(country_embeds)
        if n_cond >= 2:
            loss = eng.score_conds(xs, eps_all, t_all, n_cond, x_index=x_index, latent_dtype=self.latent_dtype, slot_table=table)
        else:
            slots = table.reshape(-1) if table is not None else torch.zeros(n * N, dtype=torch.int32)
            loss = eng.score(xs, eps_all, t_all, slots, x_index=x_index, latent_dtype=self.latent_dtype)
        self.last_loss32 = loss             # diagnostics (bench.py's score_deviation): the call's fp32 losses, rows k*n*N + j*N + i
        grid = loss.view(n_cond, n, N, *loss.shape[1:]).permute(1, 2, 0, 3, 4, 5).to(torch.float16)     # compute.py:155,160 per image
        return grid.cpu() if to_host else grid.contiguous()

    # -- D.get_path / np.save (compute.py:162-163,192) -------------------------------------------
    @staticmethod
    def get_path(typicality_path: str, path: str) -> str:
        return os.path.join(typicality_path, os.path.split(path)[1].replace(".jpg", ".npy").replace(".png", ".npy"))

    def save_grid(self, typicality_path: str, image_path: str, grid) -> str:
        out = self.get_path(typicality_path, image_path)
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        with open(out, "wb") as f:
            np.save(f, grid.cpu().numpy())
        return out

    # -- D.rescale (compute.py:165-180) ----------------------------------------------------------
    @staticmethod
    def rescale_size(which: Optional[str], width: int, height: int):
        """(width, height) `D.rescale` resizes a (width, height) image to: cars -> short side 256 with the long
        side truncated by int(); places -> short side 512 with the long side rounded up by math.ceil (a square
        image takes the `else` branch in both); every other dataset is left alone."""
        import math
        w, h = width, height
        if which == "cars":
            if w > h:
                w = int(w * 256 / h)
                h = 256
            else:
                h = int(h * 256 / w)
                w = 256
        elif which == "places":
            if width > height:
                w, h = math.ceil(width * (512 / height)), 512
            else:
                w, h = 512, math.ceil(height * (512 / width))
        return w, h

    def rescale(self, img):
        """PIL image -> PIL image, LANCZOS like the reference; identity when the size already matches the rule's
        output is NOT special-cased (the reference resamples anyway)."""
        import PIL.Image
        if self.which in ("cars", "places"):
            return img.resize(self.rescale_size(self.which, img.width, img.height), PIL.Image.LANCZOS)
        return img

    # -- D.compute / __call__ / exists (compute.py:182-202) --------------------------------------
    @torch.no_grad()
    def compute(self, country: str, path: str, vae_noise=None):
        """`D.compute(country, path)`: open + rescale the image, score it under [country, ""] and write the
        `[N,2,4,h,w]` float16 grid to `<typicality_path>/<image stem>.npy`.  Needs VAE weights on the engine
        and `country_embeds`.  Image sizes that are not multiples of 8 (cars: 256 x 341) are encoded as they are:
        every stride-2 stage of the VAE floors, as in the reference."""
        import PIL.Image
        assert self.typicality_path is not None and self.country_embeds is not None, "scorer built without D's arguments"
        img = self.rescale(PIL.Image.open(path))
        seed = os.path.split(path)[1]
        out = os.path.join(self.typicality_path, seed.replace(".jpg", ".npy").replace(".png", ".npy"))
        embeds = torch.stack([self.country_embeds[country], self.country_embeds[""]], dim=0)       # 0 = c, 1 = null
        os.makedirs(os.path.dirname(out), exist_ok=True)
        losses = self.compute_losses_from_image(self.load_image(img), embeds, vae_noise=vae_noise)
        out = self.get_path(self.typicality_path, out)
        with open(out, "wb") as f:
            np.save(f, losses.numpy())
        return out

    @torch.no_grad()
    def compute_submission(self, lines, images_per_call: int = 8, vae_noise=None):
        """`compute_submission`'s loop (compute.py:284-290) over `path,country` work-list lines (strings, or (path, country)
        pairs) — the reference calls `D.compute(country, path)` image by image; here runs of up to `images_per_call` consecutive
        images of one latent size go through ONE `compute_losses_batch` call, each image under its own [country, ""] prompts.
        Writes the same `<typicality_path>/<stem>.npy` files (bit-equal to `compute`'s: a sample's bits do not depend on the
        batch it rides in); returns their paths.  `vae_noise`: {path: posterior draw [1,4,h,w]} (optional, as `compute`'s)."""
        import PIL.Image
        assert self.typicality_path is not None and self.country_embeds is not None, "scorer built without D's arguments"
        items = [tuple(l.strip().split(",")) if isinstance(l, str) else tuple(l) for l in lines]
        out_paths, run = [], []              # run: (path, country, latent x [1,4,h,w])

        def flush():
            if not run:
                return
            xs = torch.cat([r[2] for r in run])
            emb = torch.stack([torch.stack([self.country_embeds[r[1]], self.country_embeds[""]]) for r in run])   # 0 = c, 1 = null
            grids = self.compute_losses_batch(xs, emb)                                 # [n, N, 2, 4, h, w] fp16 on the host
            for (path, _, _), g in zip(run, grids):
                out = self.get_path(self.typicality_path, path)
                os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
                with open(out, "wb") as f:
                    np.save(f, g.numpy())
                out_paths.append(out)
            run.clear()
        for path, country in items:
            img = self.rescale(PIL.Image.open(path))
            x = self.encode_vae(self.load_image(img), None if vae_noise is None else vae_noise.get(path))
            if run and (tuple(run[0][2].shape) != tuple(x.shape) or len(run) >= images_per_call):
                flush()
            run.append((path, country, x))
        flush()
        return out_paths

    def __call__(self, path: str):
        """`D.__call__`: the stored grid of an image."""
        return np.load(self.get_path(self.typicality_path, path))

    def exists(self, path: str) -> bool:
        return os.path.isfile(self.get_path(self.typicality_path, path))

    # -- consumers' reductions (xray/compute.py:210-218, cluster.py:517-531) on the GPU ----------
    def heatmap(self, grid):
        return self.engine.reduce_typicality(grid)[0]

    def typicality_scalar(self, grid):
        return self.engine.reduce_typicality(grid)[1]

    def load_typicality(self, grid, image_size, kx: int, ky: int):
        """`Cluster.load_typicality` (cluster.py:125-137) from the grid instead of the .npy path:
        bilinear resize to image_size = (H, W), kx x ky stride-1 average pooling per condition,
        -(pool(c) - pool(null)), mean over N -> [H-kx+1, W-ky+1] fp32 on the GPU.  Like the reference's
        `pool` (utils.py:74-80) the window is applied only when BOTH kx and ky differ from 1."""
        if kx == 1 or ky == 1:
            kx = ky = 1
        return self.engine.typicality_image(grid, image_size, kx, ky)

    def load_typicality_norm(self, grid, image_size):
        """`Cluster.load_typicality_norm` (cluster.py:112-123): the per-pixel map `(dm[:, 1] - dm[:, 0]).mean(0)` at the
        image size, then `normalize` (cluster.py:32-47) -> [H, W] fp32 in [0, 1] on the GPU."""
        return self.engine.normalize_map(self.engine.typicality_image(grid, image_size, 1, 1), "signed")

    def d_compute(self, grid, h: int, w: int, x_start: int, y_start: int, x_end: int, y_end: int):
        """`d_compute` (utils.py:122-134): the per-pixel map at (h, w) divided by its max |.|, cropped to the box."""
        return self.engine.normalize_map(self.engine.typicality_image(grid, (h, w), 1, 1), "maxabs")[x_start:x_end, y_start:y_end]

    def pixel_heatmap(self, grid, image_size):
        """`Typicallity.compute` dm_pixel (xray/compute.py:210-218): per-pixel E_N[L_null - L_c] at image size."""
        return self.engine.typicality_image(grid, image_size, 1, 1)


def shard_indices(n_items: int, rank: int, world: int) -> Sequence[int]:
    """Image-major sharding `subs[i::sub_split]` of compute.py:339."""
    return list(range(rank, n_items, world))


def _all_gather_padded(local: torch.Tensor, per: int, world: int) -> torch.Tensor:
    """ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) of `local` [n_local, ...] padded to `per` rows;
    returns [world, per, ...]."""
    import torch.distributed as dist
    # gloo has no all_gather for device tensors: a gloo group over GPU tensors (bench.py's one-GPU rehearsal of the N > 1 path) stages
    # the few scalars through the host; RCCL ("nccl") gathers on the device
    via_host = local.is_cuda and dist.get_backend() == "gloo"
    dev = torch.device("cpu") if via_host else local.device
    buf = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    buf[: local.shape[0]] = local.to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)                      # the same call on RCCL and on gloo (the CPU tests exercise exactly this path)
    return torch.stack(out, 0).to(local.device)


def gather_scores(local_scores: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """Single all-gather of the per-image T(x|c) scalars of the ranks' `r::world` shards; returns them in global
    image order.  No other collective exists on the path.  At world 1 this is the identity: the local tensor is
    returned as it is (no kernel, no copy)."""
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        assert local_scores.numel() == n_items, (local_scores.shape, n_items)
        return local_scores if local_scores.dtype == torch.float32 else local_scores.to(torch.float32)
    per = (n_items + world - 1) // world
    allv = _all_gather_padded(local_scores.to(torch.float32).reshape(-1), per, world)        # [world, per]
    # rank r holds items r, r + world, ...: item i sits at allv[i % world, i // world]  ->  transpose + trim
    return allv.t().reshape(-1)[:n_items].contiguous()


def gather_grids(local_grids: torch.Tensor, n_items: int, rank: int, world: int) -> torch.Tensor:
    """Second gather mode: the fp16 loss grids [n_local, N, n_cond, 4, h, w] of the ranks' `r::world`
    image shards -> [n_items, N, n_cond, 4, h, w] in global image order (one all-gather; 655 KB per image at N = 10)."""
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        assert local_grids.shape[0] == n_items
        return local_grids
    per = (n_items + world - 1) // world
    allv = _all_gather_padded(local_grids, per, world)                                        # [world, per, ...]
    return allv.transpose(0, 1).reshape((per * world,) + tuple(local_grids.shape[1:]))[:n_items].contiguous()


def draw_shard(n_draws: int, rank: int, world: int) -> Sequence[int]:
    """Draws of ONE image a rank scores when there are fewer images than ranks (SURVEY 8e fallback): `r::world`."""
    return list(range(rank, n_draws, world))


@torch.no_grad()
def compute_losses_draw_split(scorer: "TypicalityScorer", x, country_embeds, rank: int, world: int, B: int = 10,
                              noises=None, timesteps=None):
    """`D.compute_losses` of ONE image with its N draws split over the ranks (n_img < world): every rank
    makes the same N (eps, t) draws (same seed, compute.py:139-141), scores draws `rank::world` under all prompts, and
    one all-gather of the fp16 grids reassembles [N, n_cond, 4, h, w] on every rank — bit-equal to the single-rank grid,
    because a sample's loss does not depend on the batch it rides in."""
    if noises is None or timesteps is None:
        noises, timesteps = scorer.draw(x.shape)
    N = noises.shape[0]
    mine = draw_shard(N, rank, world)
    idx = torch.as_tensor(mine, dtype=torch.long)
    n_cond = country_embeds.shape[0]
    if len(mine):
        local = scorer.compute_losses(x, country_embeds, B, noises=noises[idx], timesteps=timesteps[idx], to_host=False)
    else:
        local = torch.zeros((0, n_cond) + tuple(x.shape[1:]), dtype=torch.float16, device=scorer.device)
    return gather_grids(local, N, rank, world)


@torch.no_grad()
def score_images_sharded(scorer: "TypicalityScorer", latents, country_embeds, rank: int, world: int, B: int = 10,
                         mode: str = "scalars", images_per_call: int = 8):
    """The multi-GPU scoring step: `latents` = the whole work list [n_img, 4, h, w] (every rank sees the
    list, as every reference process sees the submission files, compute.py:337-341); `country_embeds` [n_cond,77,768] (one
    prompt set) or [n_img,n_cond,77,768] (image j under its own category, compute.py:284-290).
      n_img >= world: rank r scores images `r::world`, `images_per_call` at a time through `compute_losses_batch` (one engine
                      call each); ONE all-gather of the per-image T(x|c) fp32 scalars (mode "scalars") or of the fp16 grids
                      (mode "grids");
      n_img <  world: every image's N draws are split over the ranks and the grids gathered (`compute_losses_draw_split`);
                      the scalars are then reduced from the full grids on every rank.
    Returns T(x|c) [n_img] fp32 (mode "scalars") or the grids [n_img, N, n_cond, 4, h, w] fp16 (mode "grids")."""
    assert mode in ("scalars", "grids")
    n_img = latents.shape[0]
    per_image = country_embeds.dim() == 4
    if n_img >= world:
        mine = shard_indices(n_img, rank, world)
        grids = []
        for c0 in range(0, len(mine), max(1, images_per_call)):
            idx = torch.as_tensor(mine[c0:c0 + max(1, images_per_call)], dtype=torch.long)
            emb = country_embeds[idx] if per_image else country_embeds
            grids.append(scorer.compute_losses_batch(latents[idx], emb, B, to_host=False))
        local = torch.cat(grids) if grids else None
        if mode == "grids":
            if local is None:
                local = torch.zeros((0,), dtype=torch.float16, device=scorer.device)
            return gather_grids(local, n_img, rank, world)
        local = torch.cat([scorer.typicality_scalar(g).reshape(1) for g in local]) if local is not None else \
            torch.zeros(0, dtype=torch.float32, device=scorer.device)
        return gather_scores(local, n_img, rank, world)
    grids = torch.stack([compute_losses_draw_split(scorer, latents[i:i + 1], country_embeds[i] if per_image else country_embeds,
                                                   rank, world, B) for i in range(n_img)])
    if mode == "grids":
        return grids
    return torch.cat([scorer.typicality_scalar(g).reshape(1) for g in grids])
