"""VAE encoder on the GPU: the 64-channel-wave igemm instantiation, the
head_dim-512 attention, and `dm_vae_encode` end to end against the CPU oracle (`oracle/vae_ref.py`,
parity unpinned — see its header) and the committed golden fixture.

Tolerances as for the U-Net: per-op rel-L2 <= 2e-3 vs torch fp32 on the same fp16
inputs; end-to-end moments rel-L2 <= 3e-3 (2x the measured value) against the fp16-autocast emulation of the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests import gpu_util as U  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def vae_sd():
    from diff_mining_amd import synth
    return synth.synth_vae_state_dict(seed=0, dtype=np.float16)


@pytest.fixture(scope="module")
def vae_engine(vae_sd):
    from diff_mining_amd.engine import UNetEngine
    assert torch.cuda.is_available()
    eng = UNetEngine(0)
    eng.load_vae_state_dict(vae_sd)
    yield eng
    eng.close()


@pytest.mark.parametrize("M,K,Cout", [(300, 128, 128), (257, 512, 256), (1000, 256, 1536), (64, 64, 128)])
def test_igemm64_dense(M, K, Cout):
    x = U.f16_randn(1, 1, M, K, seed=1)
    w = U.f16_randn(Cout, K, seed=2, scale=K ** -0.5)
    b = U.f16_randn(Cout, seed=3, scale=0.1)
    r = U.f16_randn(1, 1, M, Cout, seed=4)
    d = U.dev()
    ref = F.linear(x.float().view(M, K), w.float(), b.float())
    y = U.op_igemm(x.to(d), w.to(d), b.to(d))
    U.assert_close_fp16(y.view(M, Cout), ref, "dense64+bias")
    y = U.op_igemm(x.to(d), w.to(d), b.to(d), res=r.to(d))
    U.assert_close_fp16(y.view(M, Cout), ref.half().float() + r.float().view(M, Cout), "dense64+bias+res")


def test_igemm64_identity_asymmetric():
    K = Cout = 256
    w = torch.eye(K).half()
    x = (torch.arange(200 * K).view(1, 1, 200, K) % 89).half() / 16
    y = U.op_igemm(x.to(U.dev()), w.to(U.dev()))
    assert torch.equal(y.cpu().view(200, K), x.view(200, K))


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 7, 128, 128), (1, 16, 16, 128, 256), (1, 8, 12, 512, 512)])
def test_igemm64_conv3x3(N, H, W, Cin, Cout):
    x = U.f16_randn(N, Cin, H, W, seed=5)
    w = U.f16_randn(Cout, Cin, 3, 3, seed=6, scale=(9 * Cin) ** -0.5)
    b = U.f16_randn(Cout, seed=7, scale=0.1)
    res = U.f16_randn(N, Cout, H, W, seed=9)
    d = U.dev()
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    y = U.op_igemm(U.to_nhwc(x).to(d), U.pack_conv3(w).to(d), b.to(d), mode=1)
    U.assert_close_fp16(U.to_nchw(y), ref, "conv3x3 (64-ch waves)")
    y = U.op_igemm(U.to_nhwc(x).to(d), U.pack_conv3(w).to(d), b.to(d), res=U.to_nhwc(res).to(d), mode=1)
    U.assert_close_fp16(U.to_nchw(y), ref.half().float() + res.float(), "conv3x3+res (64-ch waves)")


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 128), (1, 8, 24, 256), (3, 4, 4, 512)])
def test_igemm64_downsample_pad0(N, H, W, C):
    """`Downsample2D(padding=0)`: F.pad(x, (0,1,0,1)) then conv3x3 stride 2 (mode 4)."""
    x = U.f16_randn(N, C, H, W, seed=11)
    w = U.f16_randn(C, C, 3, 3, seed=12, scale=(9 * C) ** -0.5)
    b = U.f16_randn(C, seed=13, scale=0.1)
    d = U.dev()
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2, padding=0)
    y = U.op_igemm(U.to_nhwc(x).to(d), U.pack_conv3(w).to(d), b.to(d), mode=4, OH=H // 2, OW=W // 2)
    assert tuple(U.to_nchw(y).shape) == tuple(ref.shape)
    U.assert_close_fp16(U.to_nchw(y), ref, "conv3x3 s2 pad(0,1,0,1)")


def _attn512(q, k, v):
    from diff_mining_amd import engine as E
    lib = E.load_library()
    B, T, Cc = q.shape
    qkv = torch.cat([q, k, v], dim=2).contiguous().to(U.dev())          # [B,T,1536], as the engine lays it out
    o = torch.empty(B, T, Cc, dtype=torch.float16, device=U.dev())
    rc = lib.dm_op_attention512(U.stream(), U.ptr(qkv), C.c_void_p(qkv.data_ptr() + Cc * 2),
                                C.c_void_p(qkv.data_ptr() + 2 * Cc * 2), U.ptr(o), B, T, 3 * Cc, Cc, float(Cc) ** -0.5)
    assert rc == 0
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize("B,T", [(2, 64), (1, 256), (3, 80), (1, 1344), (1, 33)])
def test_attention512(B, T):
    """softmax(Q K^T / sqrt(512)) V, one head; T covers multiples of the tiles and ragged tails."""
    q = U.f16_randn(B, T, 512, seed=21, scale=1.5)
    k = U.f16_randn(B, T, 512, seed=22, scale=1.5)
    v = U.f16_randn(B, T, 512, seed=23)
    ref = torch.softmax(torch.matmul(q.float(), k.float().transpose(1, 2)) * 512 ** -0.5, dim=-1) @ v.float()
    o = _attn512(q, k, v)
    U.assert_close_fp16(o, ref, f"attention512 T={T}", rel=2e-3, abs_frac=3e-3)


def test_attention512_peaked_rows():
    """Rows dominated by one key (large logits): exercises the running-max rescale."""
    B, T = 1, 128
    q = U.f16_randn(B, T, 512, seed=31, scale=4.0)
    k = q.clone()
    v = U.f16_randn(B, T, 512, seed=33)
    ref = torch.softmax(torch.matmul(q.float(), k.float().transpose(1, 2)) * 512 ** -0.5, dim=-1) @ v.float()
    o = _attn512(q, k, v)
    U.assert_close_fp16(o, ref, "attention512 peaked", rel=2e-3, abs_frac=3e-3)


def _oracle(vae_sd, img, noise, autocast=True):
    from oracle import vae_ref
    sd = {k: torch.from_numpy(v).float() for k, v in vae_sd.items()}
    return vae_ref.vae_encode(sd, img.float(), noise, autocast)


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 64, 96), (1, 128, 128), (1, 100, 68), (1, 85, 131)])
def test_vae_encode_matches_oracle(vae_engine, vae_sd, B, H, W):
    from diff_mining_amd import synth
    # sizes that are not multiples of 8 floor at every stride-2 stage (100 -> 50 -> 25 -> 12; 85 -> 42 -> 21 -> 10), like
    # diffusers' Downsample2D(padding=0): what D.rescale hands the VAE for cars (256 x 341, compute.py:165-173)
    img = torch.from_numpy(synth.synth_image(B, H, W))
    noise = U.f16_randn(B, 4, H // 8, W // 8, seed=41)
    lat, mom = vae_engine.vae_encode(img, noise, return_moments=True, out_dtype=torch.float32)
    assert lat.shape == (B, 4, H // 8, W // 8)
    ref_lat, ref_mom = _oracle(vae_sd, img, noise, autocast=True)
    ref32_lat, ref32_mom = _oracle(vae_sd, img, noise, autocast=False)
    r_ac, r_32 = U.rel_l2(mom, ref_mom), U.rel_l2(mom, ref32_mom)
    base = U.rel_l2(ref_mom, ref32_mom)
    print(f"vae {B}x{H}x{W}: moments rel-L2 vs autocast-oracle {r_ac:.2e}, vs fp32 {r_32:.2e}; oracle ac-vs-fp32 {base:.2e}")
    assert r_ac < 3e-3 and r_32 < 2.7e-3            # measured (r02) <= 1.53e-3 / 1.36e-3
    print(f"vae latents rel-L2 {U.rel_l2(lat, ref_lat):.2e}")
    assert U.rel_l2(lat, ref_lat) < 1.8e-3           # measured (r02) <= 9.0e-4
    # posterior mode = mean * scaling
    mode = vae_engine.vae_encode(img, None, out_dtype=torch.float32)
    assert torch.equal(mode, mom[:, :4] * np.float32(0.18215))
    # fp16 output = rounded fp32 output
    lat16 = vae_engine.vae_encode(img, noise)
    assert torch.equal(lat16, lat.half())


def test_vae_golden(vae_engine):
    g = np.load(os.path.join(GOLDEN, "vae_64x64.npz"))
    img, noise = torch.from_numpy(g["image"]), torch.from_numpy(g["noise"])
    lat, mom = vae_engine.vae_encode(img, noise, return_moments=True, out_dtype=torch.float32)
    print(f"vae golden: moments {U.rel_l2(mom, torch.from_numpy(g['moments'])):.2e} latents {U.rel_l2(lat, torch.from_numpy(g['latents'])):.2e}")
    assert U.rel_l2(mom, torch.from_numpy(g["moments"])) < 3e-3           # measured (r02) 1.50e-3
    assert U.rel_l2(lat, torch.from_numpy(g["latents"])) < 1.7e-3         # measured (r02) 8.5e-4


def test_vae_full_size_properties(vae_engine):
    """512x512 (BASELINE image size), batch 3 with a duplicated image: deterministic, batch-position
    invariant (bit-equal), finite, and the latent grid has the U-Net's input shape."""
    from diff_mining_amd import synth
    img = torch.from_numpy(synth.synth_image(2, 512, 512))
    img = torch.cat([img, img[:1]], 0)
    noise = U.f16_randn(3, 4, 64, 64, seed=43)
    noise[2] = noise[0]
    a, m = vae_engine.vae_encode(img, noise, return_moments=True)
    b = vae_engine.vae_encode(img, noise)
    assert a.shape == (3, 4, 64, 64) and torch.isfinite(m).all()
    assert torch.equal(a, b)
    assert torch.equal(a[0], a[2]) and not torch.equal(a[0], a[1])
    one = vae_engine.vae_encode(img[1:2], noise[1:2])
    assert torch.equal(one[0], a[1])


def test_vae_rejects_bad_input(vae_engine, vae_sd):
    from diff_mining_amd.engine import EngineError, UNetEngine
    with pytest.raises(EngineError):
        vae_engine.lib.dm_vae_encode.restype  # noqa: B018  (attribute exists)
        bad = torch.zeros(1, 3, 4, 64, dtype=torch.float16, device=U.dev())       # smaller than one latent pixel
        out = torch.empty(1, 4, 1, 8, dtype=torch.float16, device=U.dev())
        vae_engine._check(vae_engine.lib.dm_vae_encode(vae_engine._h, U.ptr(bad), None, 1, 1, 4, 64, 0.18215, U.ptr(out), None,
                                                       None, U.stream()), "dm_vae_encode")
    eng = UNetEngine(0)
    try:
        with pytest.raises(EngineError):          # no VAE weights loaded
            eng.vae_encode(torch.zeros(1, 3, 64, 64))
        sd = dict(vae_sd)
        sd.pop("encoder.conv_out.bias")
        with pytest.raises(EngineError):          # incomplete state dict
            eng.load_vae_state_dict(sd)
    finally:
        eng.close()


def test_vae_legacy_attention_names(vae_sd):
    """Original SD checkpoints name the mid-block attention query/key/value/proj_attn with [C,C,1,1] weights."""
    from diff_mining_amd import synth
    from diff_mining_amd.engine import UNetEngine
    legacy = {}
    m = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in vae_sd.items():
        if ".attentions.0.to_" in k:
            head, leaf = k.split(".attentions.0.")
            mod, wb = leaf.rsplit(".", 1)
            v = v.reshape(512, 512, 1, 1) if wb == "weight" else v
            legacy[f"vae.{head}.attentions.0.{m[mod]}.{wb}"] = v
        else:
            legacy["vae." + k] = v
    legacy["vae.decoder.conv_in.bias"] = np.zeros(512, np.float16)          # ignored
    img = torch.from_numpy(synth.synth_image(1, 64, 64))
    e1, e2 = UNetEngine(0), UNetEngine(0)
    try:
        e1.load_vae_state_dict(vae_sd)
        e2.load_vae_state_dict(legacy)
        assert torch.equal(e1.vae_encode(img), e2.vae_encode(img))
    finally:
        e1.close(); e2.close()


def test_image_to_typicality_grid(vae_engine, vae_sd, sd15_weights_f16):
    """Pixels -> latents -> [N,2,4,h,w] grid on one engine (compute.py:134-160 incl. :137), against the
    two oracles chained the same way."""
    from diff_mining_amd import synth
    from diff_mining_amd.typicality import TypicalityScorer
    from oracle import unet_ref as R
    from oracle import vae_ref
    eng = vae_engine
    if not eng._finalized:
        eng.load_state_dict(sd15_weights_f16)
    sc = TypicalityScorer(eng, seed=42, N=2, t_min=0.1, t_max=0.7)
    img = torch.from_numpy(synth.synth_image(1, 64, 64))
    vnoise = U.f16_randn(1, 4, 8, 8, seed=51)
    _, _, _, c = synth.synth_inputs(1, 2, 8, 8)
    c = torch.from_numpy(c)
    grid = sc.compute_losses_from_image(img, c, vae_noise=vnoise)
    assert grid.shape == (2, 2, 4, 8, 8) and grid.dtype == torch.float16
    vsd = {k: torch.from_numpy(v).float() for k, v in vae_sd.items()}
    usd = {k: torch.from_numpy(v).float() for k, v in sd15_weights_f16.items()}
    x_ref, _ = vae_ref.vae_encode(vsd, img.float(), vnoise.float(), autocast=True)
    noises, ts = sc.draw((1, 4, 8, 8))
    ref = R.compute_losses(usd, x_ref, c.float(), noises, ts, B=2)        # fp32 latent, fp32 draws: the reference's flow
    print(f"image -> grid rel-L2 {U.rel_l2(grid, ref):.2e}")
    assert U.rel_l2(grid, ref) < 3.2e-3              # measured (r02) 1.62e-3
    # uint8 image path: load_image reproduces to_tensor(x) * 2 - 1
    u8 = ((img[0].permute(1, 2, 0).float().numpy() + 1) * 127.5).round().clip(0, 255).astype(np.uint8)
    t = sc.load_image(u8)
    assert t.shape == (1, 3, 64, 64) and float(t.min()) >= -1 and float(t.max()) <= 1
    assert torch.allclose(t, torch.from_numpy(u8).permute(2, 0, 1)[None].float() / 255 * 2 - 1)


def test_vae_draws_per_image(vae_engine):
    """D posterior samples per image from one encoder pass == D separate encodes of the repeated image
    (dift.py:220,187 encodes the same image `ensemble_size` times)."""
    from diff_mining_amd import synth
    img = torch.from_numpy(synth.synth_image(2, 64, 64))
    noise = U.f16_randn(6, 4, 8, 8, seed=61)
    a, m = vae_engine.vae_encode(img, noise, draws_per_image=3, return_moments=True)
    b, m2 = vae_engine.vae_encode(img.repeat_interleave(3, 0), noise, return_moments=True)
    assert a.shape == (6, 4, 8, 8) and m.shape == (2, 8, 8, 8)
    assert torch.equal(a, b) and torch.equal(m, m2[::3])


def test_dift_from_pixels(vae_engine, vae_sd, sd15_weights_f16):
    """SDFeaturizer.forward from an image (dift.py:214-232): one encoder pass, `ensemble` posterior samples,
    DIFT tap, ensemble mean — against the two oracles chained (fp32 U-Net oracle like the reference's DIFT)."""
    from diff_mining_amd import synth
    from diff_mining_amd.dift import SDFeaturizer
    from oracle import unet_ref as R
    from oracle import vae_ref
    eng = vae_engine
    if not eng._finalized:
        eng.load_state_dict(sd15_weights_f16)
    ens = 2
    img = torch.from_numpy(synth.synth_image(1, 128, 128))
    vnoise = U.f16_randn(ens, 4, 16, 16, seed=71).float()
    noise = U.f16_randn(ens, 4, 16, 16, seed=72).float()
    _, _, _, c = synth.synth_inputs(1, 1, 16, 16)
    prompt = torch.from_numpy(c[:1])
    fz = SDFeaturizer(eng)
    got = fz.forward_image(img[0], prompt, t=161, up_ft_index=1, ensemble_size=ens, vae_noise=vnoise, noise=noise)
    assert got.shape == (1, 1280, 8, 8)
    vsd = {k: torch.from_numpy(v).float() for k, v in vae_sd.items()}
    usd = {k: torch.from_numpy(v).float() for k, v in sd15_weights_f16.items()}
    mom = vae_ref.vae_moments(vsd, img.float(), autocast=True)
    lat = vae_ref.posterior_sample(mom.repeat(ens, 1, 1, 1), vnoise)
    noisy = R.add_noise(lat, noise, torch.tensor(161))
    ft, _ = R.dift_features(usd, noisy.half().float(), 161, prompt.float().expand(ens, -1, -1), 1)
    ref = ft.mean(0, keepdim=True)
    print(f"dift from pixels rel-L2 {U.rel_l2(got, ref):.2e}")
    assert U.rel_l2(got, ref) < 3e-3, U.rel_l2(got, ref)      # measured (r02) 1.48e-3


def test_compute_writes_the_reference_npy(vae_engine, sd15_weights_f16, tmp_path):
    """`D.compute(country, path)` end to end (compute.py:182-192): image file -> rescale -> VAE -> N x 2 U-Net scorings ->
    `<typicality_path>/<stem>.npy` ([N,2,4,h,w] float16, cond 0 = country, 1 = ""), then `exists` / `__call__`."""
    import PIL.Image
    from diff_mining_amd import synth
    from diff_mining_amd.typicality import TypicalityScorer
    eng = vae_engine
    if not eng._finalized:
        eng.load_state_dict(sd15_weights_f16)
    _, _, _, c = synth.synth_inputs(1, 1, 8, 8)
    embeds = {"1970": torch.from_numpy(c[0]), "": torch.from_numpy(c[1])}
    out_dir = str(tmp_path / "typ" / "1970")
    sc = TypicalityScorer(eng, seed=42, N=2, t_min=0.1, t_max=0.7, typicality_path=out_dir, which="cars", country_embeds=embeds)
    u8 = ((synth.synth_image(1, 50, 67)[0].transpose(1, 2, 0).astype(np.float32) + 1) * 127.5).round().clip(0, 255).astype(np.uint8)
    path = str(tmp_path / "1970__car_000123.jpg")
    PIL.Image.fromarray(u8).save(path, quality=95)
    assert not sc.exists(path)
    vnoise = U.f16_randn(1, 4, 32, 42, seed=77)            # cars: 67 x 50 px -> (343, 256) -> latent 32 x 42
    out = sc.compute("1970", path, vae_noise=vnoise)
    assert out == os.path.join(out_dir, "1970__car_000123.npy") and sc.exists(path)
    grid = sc(path)
    assert grid.dtype == np.float16 and grid.shape == (2, 2, 4, 32, 42)
    # the same image through the pieces
    img = sc.rescale(PIL.Image.open(path))
    assert img.size == (343, 256)
    ref = sc.compute_losses_from_image(sc.load_image(img), torch.stack([embeds["1970"], embeds[""]]), vae_noise=vnoise)
    assert np.array_equal(grid, ref.numpy())
    assert np.isfinite(grid.astype(np.float32)).all() and not np.array_equal(grid[:, 0], grid[:, 1])


def test_compute_submission_writes_what_compute_writes(vae_engine, sd15_weights_f16, tmp_path):
    """`compute_submission` (compute.py:284-290) over a work list of `path,country` lines with two categories and two image
    sizes: runs of same-size images go through ONE `compute_losses_batch` call each, and every `.npy` is bit-equal to the file
    `D.compute(country, path)` writes for that image alone."""
    import PIL.Image
    from diff_mining_amd import synth
    from diff_mining_amd.typicality import TypicalityScorer
    eng = vae_engine
    if not eng._finalized:
        eng.load_state_dict(sd15_weights_f16)
    _, _, _, c = synth.synth_inputs(1, 1, 8, 8)
    g = torch.Generator().manual_seed(3)
    embeds = {"1970": torch.from_numpy(c[0]), "1985": torch.randn(77, 768, generator=g).half(), "": torch.from_numpy(c[1])}
    imgs = synth.synth_image(4, 64, 64)
    work, vnoise = [], {}
    for i, (country, hw) in enumerate([("1970", (64, 64)), ("1985", (64, 64)), ("1970", (64, 64)), ("1985", (48, 64))]):
        u8 = ((imgs[i][:, :hw[0], :hw[1]].transpose(1, 2, 0).astype(np.float32) + 1) * 127.5).round().clip(0, 255).astype(np.uint8)
        path = str(tmp_path / f"{country}__img_{i:03d}.png")
        PIL.Image.fromarray(u8).save(path)
        work.append(f"{path},{country}")
        vnoise[path] = U.f16_randn(1, 4, hw[0] // 8, hw[1] // 8, seed=90 + i)
    a = TypicalityScorer(eng, seed=42, N=3, t_min=0.1, t_max=0.7, typicality_path=str(tmp_path / "batched"), which="geo", country_embeds=embeds)
    b = TypicalityScorer(eng, seed=42, N=3, t_min=0.1, t_max=0.7, typicality_path=str(tmp_path / "single"), which="geo", country_embeds=embeds)
    outs = a.compute_submission(work, images_per_call=8, vae_noise=vnoise)
    assert len(outs) == 4
    for line in work:
        path, country = line.split(",")
        b.compute(country, path, vae_noise=vnoise[path])
        ga, gb = a(path), b(path)
        assert ga.dtype == np.float16 and ga.shape == gb.shape and ga.shape[:3] == (3, 2, 4)
        assert np.array_equal(ga, gb), f"{os.path.basename(path)}: the batched work list wrote a different grid"
