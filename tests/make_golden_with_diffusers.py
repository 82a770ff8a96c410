"""Pins the oracle (and through it the engine) to the REAL dependency of the reference's hot path.

    pip install "diffusers==0.24.0" "transformers>=4.30" "accelerate"     # the reference's pin (environment.yaml:15); torch as installed
    python tests/make_golden_with_diffusers.py             # writes tests/golden/{score,dift,vae}_diffusers.npz (CPU: ~2 min; a GPU adds the autocast arrays)
    python tests/make_golden_with_diffusers.py --check     # validates existing fixtures against the keys / shapes / dtypes the tests read
                                                           # (needs neither diffusers nor a GPU: it only opens the .npz files)

NOT runnable in the build image (diffusers is absent there and there is no network — SURVEY.md §0 F4), and never
imported by the product, the GPU tests or bench.py: it only WRITES fixtures.  Anyone with the reference's
environment runs it once and commits the files; from then on
    tests/test_oracle.py::test_oracle_against_real_diffusers_fixture      (CPU tier)
    tests/test_gpu_e2e.py::test_against_real_diffusers_fixture            (GPU tier)
compare the oracle / the engine with diffusers' own outputs and parity stops being "unpinned".

What it does (the reference's call sequence, not a copy of its code):
  * `UNet2DConditionModel(**SD15_UNET_CONFIG)` — the public `unet/config.json` of runwayml/stable-diffusion-v1-5 —
    loaded (strict) with the deterministic synthetic weights of diff-mining_amd/synth.py (no checkpoint needed);
  * `PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)`,
    the scheduler of the stock pipeline the reference loads (compute.py:65-70);
  * the three calls of `SD.compute_loss` (compute.py:99-101): scheduler.add_noise -> unet(...).sample -> F.mse_loss
    (reduction="none"), in fp32 on the CPU and — when a GPU is present — under torch.autocast(float16) exactly as
    compute.py:98 does, with the fp32 latents / draws the reference's `encode_vae` / `randn_like` produce;
  * optionally the DIFT tap (dift.py:24-169 exits after up_blocks[1]; reproduced here with a forward hook on the
    stock model's `up_blocks[1]`, which sees the same tensor) and `AutoencoderKL.encode` for the VAE oracle.
Outputs: tests/golden/score_diffusers.npz, dift_diffusers.npz, vae_diffusers.npz (inputs + outputs + versions).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
)
SD15_VAE_CONFIG = dict(
    in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
    block_out_channels=(128, 256, 512, 512), layers_per_block=2, act_fn="silu", latent_channels=4, norm_num_groups=32,
    sample_size=512, scaling_factor=0.18215,
)


def _tile(eps, t, c):
    n_cond, N = c.shape[0], eps.shape[0]
    return (torch.cat([eps] * n_cond), torch.cat([t] * n_cond),
            torch.cat([c[k:k + 1].expand(N, -1, -1) for k in range(n_cond)]))


# What the consumers of the fixtures read (tests/test_oracle.py::test_oracle_against_real_diffusers_fixture,
# tests/test_gpu_e2e.py::test_against_real_diffusers_fixture, tests/test_gpu_f32.py::test_fp32_net_against_real_diffusers_fixture):
# key -> (dtype kind, shape as a function of the case).  `--check` holds a fixture to this table, so whoever generates it learns
# at once — without a GPU and without the test-suite — whether the files will be picked up.
SCORE_CASES = ((8, 8, 2), (16, 16, 1), (12, 10, 1), (32, 42, 1), (32, 48, 1))        # (h, w, draws) as written below


def expected_score_keys():
    exp = {"diffusers_version": ("U", None)}
    for (h, w, n) in SCORE_CASES:
        tag, B = f"{h}x{w}", 2 * n
        exp.update({f"x_{tag}": ("f4", (1, 4, h, w)), f"eps_{tag}": ("f4", (n, 4, h, w)), f"t_{tag}": ("i8", (n,)),
                    f"c_{tag}": ("f", (2, 77, 768)), f"noisy_fp32_cpu_{tag}": ("f4", (B, 4, h, w)),
                    f"pred_fp32_cpu_{tag}": ("f4", (B, 4, h, w)), f"loss_fp32_cpu_{tag}": ("f4", (B, 4, h, w))})
    exp.update({"x": ("f4", (1, 4, 8, 8)), "eps": ("f4", (2, 4, 8, 8)), "t": ("i8", (2,)), "c": ("f", (2, 77, 768)),
                "loss_fp32_cpu": ("f4", (4, 4, 8, 8)), "pred_fp32_cpu": ("f4", (4, 4, 8, 8))})
    return exp


OPTIONAL_SCORE_KEYS = ("loss_autocast_cuda", "pred_autocast_cuda", "noisy_dtype")          # written only where a GPU ran the autocast leg


def expected_dift_keys():
    return {"diffusers_version": ("U", None), "noisy": ("f4", (2, 4, 16, 16)), "t": ("i8", ()), "prompt": ("f", (1, 77, 768)),
            "feat_fp32": ("f2", (2, 1280, 8, 8)), "feat_f32_full": ("f4", (2, 1280, 8, 8)),
            "noisy_12x10": ("f4", (2, 4, 12, 10)), "feat_fp32_12x10": ("f2", (2, 1280, 6, 5)), "feat_f32_full_12x10": ("f4", (2, 1280, 6, 5))}


def expected_vae_keys():
    return {"diffusers_version": ("U", None), "image": ("f2", (2, 3, 64, 64)), "moments": ("f4", (2, 8, 8, 8))}


def check_fixture(path, expected, optional_prefixes=()):
    """Problems of one fixture file as a list of strings (empty = the tests will read it)."""
    if not os.path.exists(path):
        return [f"{os.path.basename(path)}: absent"]
    g = np.load(path)
    bad = []
    for k, (kind, shape) in expected.items():
        if k not in g:
            bad.append(f"{os.path.basename(path)}: key {k!r} missing")
            continue
        a = g[k]
        dt = a.dtype.kind + (str(a.dtype.itemsize) if a.dtype.kind in "fi" else "")
        if kind == "U":
            if a.dtype.kind != "U":
                bad.append(f"{os.path.basename(path)}: {k} should be a string, is {a.dtype}")
            continue
        if not (dt == kind or (kind == "f" and a.dtype.kind == "f")):
            bad.append(f"{os.path.basename(path)}: {k} dtype {a.dtype}, expected {kind}")
        if shape is not None and tuple(a.shape) != tuple(shape):
            bad.append(f"{os.path.basename(path)}: {k} shape {tuple(a.shape)}, expected {tuple(shape)}")
        if a.dtype.kind == "f" and not np.isfinite(a).all():
            bad.append(f"{os.path.basename(path)}: {k} holds non-finite values")
    known = set(expected)
    for k in g.files:
        if k not in known and not any(k == o or k.startswith(o + "_") for o in optional_prefixes):
            bad.append(f"{os.path.basename(path)}: unexpected key {k!r} (a fixture is data the tests read; nothing else belongs in it)")
    if "diffusers_version" in g and "0.24" not in str(g["diffusers_version"]):
        bad.append(f"{os.path.basename(path)}: written with {g['diffusers_version']} — the reference pins diffusers==0.24.0 (environment.yaml:15)")
    return bad


def check(out_dir=OUT):
    bad = check_fixture(os.path.join(out_dir, "score_diffusers.npz"), expected_score_keys(), OPTIONAL_SCORE_KEYS)
    bad += check_fixture(os.path.join(out_dir, "dift_diffusers.npz"), expected_dift_keys())
    bad += check_fixture(os.path.join(out_dir, "vae_diffusers.npz"), expected_vae_keys())
    return bad


def main():
    if "--check" in sys.argv:
        bad = check()
        for b in bad:
            print("PROBLEM:", b)
        print("fixtures OK: the three *_diffusers tests will run" if not bad else f"{len(bad)} problem(s)")
        sys.exit(1 if bad else 0)
    try:
        import diffusers
        from diffusers import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
    except ImportError as e:                                   # the build image lands here
        sys.exit(f"diffusers is not installed ({e}); install diffusers==0.24.0 (the reference's pin) and re-run")
    from diff_mining_amd import synth
    os.makedirs(OUT, exist_ok=True)
    ver = np.array(f"diffusers {diffusers.__version__}, torch {torch.__version__}")

    # ---- scoring: SD.compute_loss on identical (x, eps, t, c) -------------------------------------------------
    unet = UNet2DConditionModel(**SD15_UNET_CONFIG).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(seed=0, dtype=np.float32).items()}
    unet.load_state_dict(sd, strict=True)                      # 686 tensors, diffusers names: must load unmodified
    sched = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                          skip_prk_steps=True)
    out = {}
    # 8x8 / 16x16: the keys the tests read; 12x10 and 32x42 (cars: 256 x 341 px -> latent 32 x 42 -> 16x21 -> 8x11 -> 4x6): odd
    # sizes, where the up path must honour `upsample_size` (dift.py:54-56,146-147); 32x48: the widest cars latent
    for (h, w, n_draws) in ((8, 8, 2), (16, 16, 1), (12, 10, 1), (32, 42, 1), (32, 48, 1)):
        x, eps, t, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, n_draws, h, w, latent_dtype=np.float32))
        nb, tb, cc = _tile(eps, t, c)
        with torch.no_grad():
            noisy = sched.add_noise(x.expand(nb.shape[0], -1, -1, -1), nb, tb)
            pred = unet(noisy, tb, cc.float()).sample
            loss32 = F.mse_loss(pred.float(), nb, reduction="none")
        tag = f"{h}x{w}"
        out.update({f"x_{tag}": x.numpy(), f"eps_{tag}": eps.numpy(), f"t_{tag}": t.numpy(), f"c_{tag}": c.numpy(),
                    f"noisy_fp32_cpu_{tag}": noisy.numpy(), f"pred_fp32_cpu_{tag}": pred.numpy(),
                    f"loss_fp32_cpu_{tag}": loss32.numpy()})
        if torch.cuda.is_available():
            dev = torch.device("cuda")
            u16 = UNet2DConditionModel(**SD15_UNET_CONFIG).eval()
            u16.load_state_dict(sd, strict=True)
            u16 = u16.to(dev, torch.float16)                   # torch_dtype=torch.float16 (compute.py:69)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                xe, ne, te = x.to(dev).expand(nb.shape[0], -1, -1, -1), nb.to(dev), tb.to(dev)
                noisy = sched.add_noise(xe, ne, te)
                pred = u16(noisy, te, cc.to(dev)).sample
                loss = F.mse_loss(pred.float(), ne, reduction="none")
            out.update({f"loss_autocast_cuda_{tag}": loss.float().cpu().numpy(),
                        f"pred_autocast_cuda_{tag}": pred.float().cpu().numpy(),
                        f"noisy_dtype_{tag}": np.array(str(noisy.dtype))})
    # the keys the tests read (8x8 case)
    for k in ("x", "eps", "t", "c", "loss_fp32_cpu", "pred_fp32_cpu", "loss_autocast_cuda"):
        if f"{k}_8x8" in out:
            out[k] = out[f"{k}_8x8"]
    np.savez_compressed(os.path.join(OUT, "score_diffusers.npz"), diffusers_version=ver, **out)

    # ---- DIFT tap: output of up_blocks[1] (dift.py:134-165), fp32 like the reference's DIFT pipeline ---------
    x, eps, _, c = (torch.from_numpy(a) if isinstance(a, np.ndarray) else a
                    for a in synth.synth_inputs(1, 2, 16, 16, latent_dtype=np.float32))
    tt = torch.tensor([161, 161])
    grabbed = {}
    hook = unet.up_blocks[1].register_forward_hook(lambda m, i, o: grabbed.__setitem__("ft", o))
    with torch.no_grad():
        noisy = sched.add_noise(x.expand(2, -1, -1, -1), eps, tt)
        unet(noisy, tt, c[:1].float().expand(2, -1, -1))
    hook.remove()
    dift = dict(noisy=noisy.numpy(), t=np.int64(161), prompt=c[:1].numpy(), feat_fp32=grabbed["ft"].numpy().astype(np.float16),
                feat_f32_full=grabbed["ft"].numpy())           # unrounded: what the fp32 net (dm_f32_dift) is held to (r04)
    # the same tap on an odd latent (12 x 10 -> 6 x 5 -> 3 x 3 -> 2 x 2; up_blocks[1] ends at 6 x 5 through upsample_size)
    xo, eo, _, _ = (torch.from_numpy(a) for a in synth.synth_inputs(1, 2, 12, 10, latent_dtype=np.float32))
    hook = unet.up_blocks[1].register_forward_hook(lambda m, i, o: grabbed.__setitem__("ft_odd", o))
    with torch.no_grad():
        noisy_o = sched.add_noise(xo.expand(2, -1, -1, -1), eo, tt)
        unet(noisy_o, tt, c[:1].float().expand(2, -1, -1))
    hook.remove()
    dift.update(noisy_12x10=noisy_o.numpy(), feat_fp32_12x10=grabbed["ft_odd"].numpy().astype(np.float16),
                feat_f32_full_12x10=grabbed["ft_odd"].numpy())
    np.savez_compressed(os.path.join(OUT, "dift_diffusers.npz"), diffusers_version=ver, **dift)

    # ---- VAE encoder moments (compute.py:91-93) ----------------------------------------------------------------
    vae = AutoencoderKL(**SD15_VAE_CONFIG).eval()
    vsd = {k: torch.from_numpy(v) for k, v in synth.synth_vae_state_dict(seed=0, dtype=np.float32).items()}
    own = vae.state_dict()
    own.update({k: v for k, v in vsd.items() if k in own})     # encoder.* and quant_conv.*; the decoder keeps its init
    vae.load_state_dict(own, strict=True)
    img = torch.from_numpy(synth.synth_image(2, 64, 64)).float()
    with torch.no_grad():
        post = vae.encode(img).latent_dist
    np.savez_compressed(os.path.join(OUT, "vae_diffusers.npz"), diffusers_version=ver, image=img.numpy().astype(np.float16),
                        moments=torch.cat([post.mean, post.logvar], 1).numpy())
    for f in sorted(os.listdir(OUT)):
        if "diffusers" in f:
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
