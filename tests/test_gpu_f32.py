"""Parity of the fp32 U-Net (dm_f32_*: the arithmetic of the reference's DIFT featuriser, dift.py:191,197-199 — no torch_dtype,
no autocast) against the fp32 CPU oracle and against plain PyTorch fp32 operators, through the C ABI.

Tolerances: both sides compute in fp32 and differ only in the ORDER of their fp32 partial sums (MFMA k order vs MKL / oneDNN
blocking).  The oracle against itself under such re-orderings moves by 2-3e-6 rel-L2 on the full U-Net
(tests/test_oracle.py::test_oracle_noise_floor_under_summation_order: 2.8-3.1e-6 on eps_hat); asserted here: operators 2e-6,
end to end 2e-5 (one decade above the floor; a single fp16 rounding anywhere in the path would give ~3e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from diff_mining_amd import engine as E  # noqa: E402
from diff_mining_amd import synth  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402

TOL_OP = 2e-6
TOL_E2E = 2e-5


@pytest.fixture(scope="module")
def net32(sd15_weights_f16):
    from diff_mining_amd.engine import UNetEngineF32
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    e = UNetEngineF32(0)
    e.load_state_dict(sd15_weights_f16)        # fp16-valued weights widened exactly: the oracle gets the same values as fp32
    yield e
    e.close()


def _randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _gemm(X, W, bias=None, X2=None, temb=None, res=None, mode=0, OH=None, OW=None):
    lib = E.load_library()
    N, H, Wd, C1 = X.shape
    Cin = C1 + (X2.shape[3] if X2 is not None else 0)
    Cout = W.shape[0]
    OH = H if OH is None else OH
    OW = Wd if OW is None else OW
    Y = torch.empty(N, OH, OW, Cout, dtype=torch.float32, device=X.device)
    rc = lib.dm_f32_op_gemm(U.stream(), U.ptr(X), U.ptr(X2), U.ptr(W), U.ptr(bias), U.ptr(temb), U.ptr(res), U.ptr(Y),
                            N, H, Wd, OH, OW, Cin, C1, Cout, mode, temb.stride(0) if temb is not None else 0)
    assert rc == 0, "dm_f32_op_gemm failed"
    torch.cuda.synchronize()
    return Y


@pytest.mark.parametrize("mode,N,H,W,C1,C2,Cout,OH,OW", [
    (1, 2, 16, 16, 320, 0, 320, 16, 16),        # ResNet conv1 / conv2
    (1, 3, 9, 7, 640, 320, 640, 9, 7),          # concat source (up block), odd image, rows not a multiple of the tile
    (2, 2, 16, 16, 320, 0, 320, 8, 8),          # Downsample2D.conv
    (2, 1, 9, 7, 320, 0, 320, 5, 4),            # stride 2 on an odd image
    (3, 2, 8, 8, 640, 0, 640, 16, 16),          # Upsample2D: nearest 2x + conv
    (3, 1, 5, 4, 320, 0, 320, 9, 7),            # nearest to a given size (upsample_size, dift.py:54-56)
    (4, 1, 16, 16, 128, 0, 128, 8, 8),          # VAE Downsample2D: pad (0,1,0,1), stride 2; Cout not a multiple of the channel tile
])
def test_gemm32_conv_modes(mode, N, H, W, C1, C2, Cout, OH, OW):
    Cin = C1 + C2
    x = _randn(N, Cin, H, W, seed=1)
    w = _randn(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = _randn(Cout, seed=3)
    temb = _randn(N, Cout + 64, seed=4) if mode == 1 else None
    if mode == 1:
        ref = F.conv2d(x, w, b, padding=1) + temb[:, 32:32 + Cout, None, None]
    elif mode == 2:
        ref = F.conv2d(x, w, b, stride=2, padding=1)
    elif mode == 3:
        ref = F.conv2d(F.interpolate(x, size=(OH, OW), mode="nearest"), w, b, padding=1)
    else:
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    assert ref.shape == (N, Cout, OH, OW)
    res = _randn(N, Cout, OH, OW, seed=5)
    ref = ref + res
    xn = U.to_nhwc(x).cuda()
    X, X2 = (xn[..., :C1].contiguous(), xn[..., C1:].contiguous()) if C2 else (xn, None)
    tb = temb.cuda() if temb is not None else None
    Y = _gemm(X, U.pack_conv3(w).cuda(), b.cuda(), X2, tb[:, 32:] if tb is not None else None, U.to_nhwc(res).cuda(), mode, OH, OW)
    r = U.rel_l2(U.to_nchw(Y).cpu(), ref)
    print(f"gemm32 mode {mode} {N}x{H}x{W} {Cin}->{Cout}: rel-L2 {r:.2e}")
    assert r < TOL_OP


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 8, 8, 640, 640), (3, 5, 7, 64, 320), (1, 1, 2, 32, 128), (2, 16, 16, 1280, 1280)])
def test_gemm32_upconv_folded(N, H, W, Cin, Cout):
    """gemm32 mode 5 (option up_fold in the fp32 net): Upsample2D — nearest 2x, then conv3x3 pad 1 — as four 2x2 convolutions on the
    source grid with the 3x3 taps that read the same source pixel pre-summed (double sum, one fp32 rounding), against
    F.conv2d(F.interpolate(x)) in fp32 at the operator tolerance; ragged rows, a one-row image, Cout below the channel tile."""
    lib = E.load_library()
    x = _randn(N, Cin, H, W, seed=1)
    w = _randn(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = _randn(Cout, seed=3)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    sel = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    wd = w.double()
    w4 = torch.stack([torch.cat([wd[:, :, sel[py][a], :][:, :, :, sel[px][bb]].sum(dim=(2, 3)) for a in (0, 1) for bb in (0, 1)], dim=1)
                      for py in (0, 1) for px in (0, 1)], 0).float().contiguous()        # [4][Cout][(a*2+b)*Cin + ci]
    X = U.to_nhwc(x).cuda()
    Y = torch.empty(N, 2 * H, 2 * W, Cout, dtype=torch.float32, device=X.device)
    W4, B = w4.cuda(), b.cuda()
    rc = lib.dm_f32_op_gemm(U.stream(), U.ptr(X), None, U.ptr(W4), U.ptr(B), None, None, U.ptr(Y), N, H, W, H, W, Cin, Cin, Cout, 5, 0)
    assert rc == 0
    torch.cuda.synchronize()
    r = U.rel_l2(U.to_nchw(Y).cpu(), ref)
    print(f"gemm32 mode 5 (folded up-sampler) {N}x{H}x{W} {Cin}->{Cout}: rel-L2 {r:.2e}")
    assert r < TOL_OP


@pytest.mark.parametrize("M,K,Nn", [(77 * 3, 768, 640), (5, 320, 1280), (1000, 1280, 20160), (4096, 320, 2560)])
def test_gemm32_dense(M, K, Nn):
    x = _randn(M, K, seed=1)
    w = _randn(Nn, K, seed=2, scale=K ** -0.5)
    b = _randn(Nn, seed=3)
    ref = F.linear(x, w, b)
    Y = _gemm(x.view(1, 1, M, K).cuda(), w.cuda(), b.cuda())
    r = U.rel_l2(Y.view(M, Nn).cpu(), ref)
    print(f"gemm32 dense {M}x{K}x{Nn}: rel-L2 {r:.2e}")
    assert r < TOL_OP


@pytest.mark.parametrize("B,heads,Tq,Tk,D,cross", [
    (2, 8, 256, 256, 40, False), (1, 8, 1024, 1024, 80, False), (2, 8, 64, 64, 160, False), (1, 8, 90, 90, 40, False),
    (3, 8, 256, 77, 40, True), (3, 8, 100, 77, 160, True),
])
def test_attention32(B, heads, Tq, Tk, D, cross):
    lib = E.load_library()
    Cc = heads * D
    nk = 2 if cross else B
    q, k, v = _randn(B, Tq, Cc, seed=1), _randn(nk, Tk, Cc, seed=2), _randn(nk, Tk, Cc, seed=3)
    slots = torch.tensor([1, 0, 1][:B], dtype=torch.int32) if cross else None
    kk, vv = (k[slots.long()], v[slots.long()]) if cross else (k, v)
    ref = F.scaled_dot_product_attention(q.view(B, Tq, heads, D).transpose(1, 2), kk.view(B, Tk, heads, D).transpose(1, 2),
                                         vv.view(B, Tk, heads, D).transpose(1, 2)).transpose(1, 2).reshape(B, Tq, Cc)
    Q, K, V = q.cuda(), k.cuda(), v.cuda()
    O = torch.empty_like(Q)
    sl = slots.cuda() if cross else None
    rc = lib.dm_f32_op_attention(U.stream(), U.ptr(Q), U.ptr(K), U.ptr(V), U.ptr(O), Cc, Cc, Cc, Cc, Tq * Cc, Tk * Cc, Tk * Cc, Tq * Cc,
                                 U.ptr(sl), nk, B, heads, Tq, Tk, D, float(D) ** -0.5)
    assert rc == 0
    torch.cuda.synchronize()
    r = U.rel_l2(O.cpu(), ref)
    print(f"attention32 B{B} Tq{Tq} Tk{Tk} D{D}: rel-L2 {r:.2e}")
    assert r < TOL_OP


def test_norms32():
    lib = E.load_library()
    N, H, W, C1, C2, G = 2, 9, 7, 1280, 640, 32
    C = C1 + C2
    x = _randn(N, C, H, W, seed=1) * 3 + 5.0        # |mean| / std ~ 1.7: the statistics are taken in fp64
    g, b = _randn(C, seed=2), _randn(C, seed=3)
    ref = F.silu(F.group_norm(x, G, g, b, 1e-5))
    xn = U.to_nhwc(x).cuda()
    Y = torch.empty(N, H, W, C, dtype=torch.float32, device="cuda")
    work = torch.empty(N * G * 2, dtype=torch.float32, device="cuda")
    gg, bb = g.cuda(), b.cuda()
    xa, xb = xn[..., :C1].contiguous(), xn[..., C1:].contiguous()
    rc = lib.dm_f32_op_groupnorm(U.stream(), U.ptr(xa), U.ptr(xb), N, H * W, C, C1, G, 1e-5,
                                 U.ptr(gg), U.ptr(bb), 1, U.ptr(work), U.ptr(Y))
    assert rc == 0
    torch.cuda.synchronize()
    r = U.rel_l2(U.to_nchw(Y).cpu(), ref)
    print(f"groupnorm32 + SiLU (concat source): rel-L2 {r:.2e}")
    assert r < TOL_OP
    rows, Cl = 333, 640
    t = _randn(rows, Cl, seed=4) * 2 + 1.0
    gl, bl = _randn(Cl, seed=5), _randn(Cl, seed=6)
    ref = F.layer_norm(t, (Cl,), gl, bl, 1e-5)
    T, Yl = t.cuda(), torch.empty(rows, Cl, dtype=torch.float32, device="cuda")
    glc, blc = gl.cuda(), bl.cuda()
    assert lib.dm_f32_op_layernorm(U.stream(), U.ptr(T), rows, Cl, U.ptr(glc), U.ptr(blc), 1e-5, U.ptr(Yl)) == 0
    torch.cuda.synchronize()
    r = U.rel_l2(Yl.cpu(), ref)
    print(f"layernorm32: rel-L2 {r:.2e}")
    assert r < TOL_OP


def _inputs(h, w, B, n_prompts=2):
    x, eps, t, c = synth.synth_inputs(1, B, h, w, latent_dtype=np.float32)
    x, eps, t, c = (torch.from_numpy(a) for a in (x, eps, t, c))
    noisy = R.add_noise(x.float().expand(B, -1, -1, -1), eps.float(), t, R.alphas_cumprod())
    return noisy, t, c[:n_prompts].float()


@pytest.mark.parametrize("h,w,B", [(8, 8, 3), (16, 16, 2), (12, 10, 2)])
def test_unet32_vs_fp32_oracle(net32, sd15_weights_torch, h, w, B):
    """`unet(sample, t, ctx).sample` in fp32 (per-sample timesteps, two prompts, an odd latent with `upsample_size`)."""
    noisy, t, c = _inputs(h, w, B)
    slots = torch.tensor([0, 1, 0][:B], dtype=torch.int32)
    net32.set_prompts(c)
    out = net32.unet(noisy, t, slots).cpu()
    ref = R.unet_forward(sd15_weights_torch, noisy, t, c[slots.long()], autocast=False)
    r = U.rel_l2(out, ref)
    mx = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"unet32 {h}x{w} B{B}: eps_hat rel-L2 vs fp32 oracle {r:.2e}, max |d| / max |ref| {mx:.2e}")
    assert out.shape == ref.shape and torch.isfinite(out).all() and r < TOL_E2E


def test_unet32_up_fold_option(net32, sd15_weights_torch):
    """Option "up_fold" in the fp32 net (gemm32 mode 5): on and off agree at the fp32 summation-order level and are equally far from
    the fp32 oracle (the summed taps are rounded at 2^-24)."""
    lib = net32.lib
    noisy, t, c = _inputs(16, 16, 2)
    slots = torch.tensor([0, 1], dtype=torch.int32)
    net32.set_prompts(c)
    out = {}
    try:
        for v in (1, 0):
            assert lib.dm_set_option(b"up_fold", v) == 0
            out[v] = net32.unet(noisy, t, slots).cpu()
    finally:
        lib.dm_set_option(b"up_fold", 1)
    ref = R.unet_forward(sd15_weights_torch, noisy, t, c[slots.long()], autocast=False)
    d, r1, r0 = U.rel_l2(out[1], out[0]), U.rel_l2(out[1], ref), U.rel_l2(out[0], ref)
    print(f"unet32 up_fold on vs off {d:.2e}; vs the fp32 oracle on {r1:.2e} off {r0:.2e}")
    # on vs off: a different k walk of three layers = a re-ordering of fp32 partial sums (the oracle against itself under re-orderings: 2.8-3.1e-6)
    assert not torch.equal(out[1], out[0]) and d < 4e-6 and r1 < TOL_E2E and r0 < TOL_E2E and r1 <= 1.2 * r0


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_dift32_vs_fp32_oracle(net32, sd15_weights_torch, idx):
    """MyUNet2DConditionModel.forward's tap at every up_ft_index + the ensemble mean of SDFeaturizer.forward (dift.py:231)."""
    h, w, ens = 16, 12, 4
    noisy, _, c = _inputs(h, w, ens, 1)
    net32.set_prompts(c)
    slots = torch.zeros(ens, dtype=torch.int32)
    feat, mean = net32.dift(noisy, torch.tensor(261), slots, idx, ens)
    ft_ref, mean_ref = R.dift_features(sd15_weights_torch, noisy, 261, c.expand(ens, -1, -1), idx)
    rf, rm = U.rel_l2(feat.cpu(), ft_ref), U.rel_l2(mean.cpu(), mean_ref)
    print(f"dift32 up_ft_index {idx}: features {tuple(feat.shape)} rel-L2 {rf:.2e}, ensemble mean {rm:.2e}")
    assert feat.shape == ft_ref.shape and mean.shape == mean_ref.shape and rf < TOL_E2E and rm < TOL_E2E


def test_dift32_chunked_batch_is_bit_identical(net32):
    """A batch larger than one run (chunks of whole ensembles): the same bits as ensemble-sized calls."""
    h = w = 64
    ens, n_img = 8, 9                                   # 72 samples > the 64-sample run at a 64 x 64 latent
    g = torch.Generator().manual_seed(3)
    noisy = torch.randn(ens * n_img, 4, h, w, generator=g)
    c = torch.from_numpy(synth.synth_inputs(1, 1, 8, 8)[3][:1]).float()
    net32.set_prompts(c)
    slots = torch.zeros(ens * n_img, dtype=torch.int32)
    _, mean = net32.dift(noisy, torch.tensor(261), slots, 1, ens)
    _, m0 = net32.dift(noisy[:ens], torch.tensor(261), slots[:ens], 1, ens)
    _, m8 = net32.dift(noisy[-ens:], torch.tensor(261), slots[:ens], 1, ens)
    assert mean.shape == (n_img, 1280, 32, 32) and torch.isfinite(mean).all()
    assert torch.equal(mean[:1], m0) and torch.equal(mean[-1:], m8)


def test_sdfeaturizer_fp32_is_the_default_arithmetic(net32, sd15_weights_f16, sd15_weights_torch):
    """`SDFeaturizer.forward` (dift.py:214-232) over the fp32 net: the reference's dtype; the fp16 engine is the opt-in fast mode."""
    from diff_mining_amd.dift import SDFeaturizer
    from diff_mining_amd.engine import UNetEngine
    h = w = 16
    ens = 4
    x, eps, _, c = synth.synth_inputs(1, ens, h, w, latent_dtype=np.float32)
    lat, noise, pe = torch.from_numpy(x), torch.from_numpy(eps), torch.from_numpy(c[:1]).float()
    f32 = SDFeaturizer(net32)
    assert f32.dtype == torch.float32
    out = f32.forward(lat, pe, t=261, up_ft_index=1, ensemble_size=ens, noise=noise)
    noisy = R.add_noise(lat.expand(ens, -1, -1, -1), noise, torch.tensor(261), R.alphas_cumprod())
    _, mean_ref = R.dift_features(sd15_weights_torch, noisy, 261, pe.expand(ens, -1, -1), 1)
    r = U.rel_l2(out.cpu(), mean_ref)
    e16 = UNetEngine(0)
    e16.load_state_dict(sd15_weights_f16)
    out16 = SDFeaturizer(e16).forward(lat, pe, t=261, up_ft_index=1, ensemble_size=ens, noise=noise)
    r16 = U.rel_l2(out16.cpu(), mean_ref)
    e16.close()
    print(f"SDFeaturizer fp32 net vs fp32 oracle {r:.2e}; fp16 engine vs fp32 oracle {r16:.2e}")
    assert out.shape == (1, 1280, h // 2, w // 2) and r < TOL_E2E and r16 < 4e-3


def _grid32(net32, x, c, noises, ts):
    """D.compute_losses (compute.py:134-160) in exact fp32 on the GPU: dm_f32_score = the reference's add_noise -> U-Net -> squared error
    with no autocast -> [N,2,4,h,w] fp32 (cond 0 = c, 1 = null) — what the CPU oracle computes with autocast=False, at any size."""
    N = noises.shape[0]
    net32.set_prompts(c.float())
    loss = net32.score_conds(x, noises, ts, 2)                     # [2N,4,h,w], cond-major
    return loss.view(2, N, *loss.shape[1:]).transpose(0, 1)


def test_gpu_fp32_ground_truth_agrees_with_the_cpu_oracle(net32, sd15_weights_torch):
    """The loss grid of `_grid32` IS the fp32 oracle's (autocast=False): checked where the CPU oracle is affordable, so that the
    full-size comparison below can stand on it."""
    N, h, w = 3, 16, 16
    x, _, _, c = synth.synth_inputs(1, 1, h, w, latent_dtype=np.float32)
    x, c = torch.from_numpy(x), torch.from_numpy(c)
    noises, ts = R.draw_noise_and_timesteps((1, 4, h, w), N, 0.1, 0.7, seed=42)
    g = _grid32(net32, x, c, noises, ts).cpu()
    # (R.compute_losses returns the reference's fp16 grid; the unrounded fp32 losses come from compute_loss on the tiled batch)
    cc = torch.cat([c[k:k + 1].float().expand(N, -1, -1) for k in range(2)])
    ref = R.compute_loss(sd15_weights_torch, x, torch.cat([noises] * 2), torch.cat([ts] * 2), cc, autocast=False)
    ref = ref.view(2, N, 4, h, w).transpose(0, 1)
    r = U.rel_l2(g, ref)
    dT = abs(R.typicality_scalar(g).item() - R.typicality_scalar(ref).item()) / ref.mean().item()
    print(f"fp32 grid on the GPU vs the fp32 CPU oracle @16x16 N=3: rel-L2 {r:.2e}, |dT|/mean loss {dT:.2e}")
    assert g.shape == ref.shape and r < TOL_E2E and dT < 1e-6


def test_gpu_fp32_ground_truth_agrees_with_the_cpu_oracle_at_the_baseline_size(net32, sd15_weights_torch):
    """`bench.py`'s `score_deviation` measures the fp16 engine against the fp32 net at 64 x 64; the fp32 net was
    tied to the CPU oracle at 16 x 16 only.  Here one draw x 2 prompts at the BASELINE latent size itself (two 803-GFLOP forwards of
    the CPU oracle, a few seconds): the GPU's exact-fp32 grid is the oracle's `autocast=False` grid at fp32 round-off."""
    N, h, w = 1, 64, 64
    x, _, _, c = synth.synth_inputs(1, 1, h, w, latent_dtype=np.float32)
    x, c = torch.from_numpy(x), torch.from_numpy(c)
    noises, ts = R.draw_noise_and_timesteps((1, 4, h, w), N, 0.1, 0.7, seed=42)
    g = _grid32(net32, x, c, noises, ts).cpu()
    cc = torch.cat([c[k:k + 1].float().expand(N, -1, -1) for k in range(2)])
    ref = R.compute_loss(sd15_weights_torch, x, torch.cat([noises] * 2), torch.cat([ts] * 2), cc, autocast=False)
    ref = ref.view(2, N, 4, h, w).transpose(0, 1)
    r = U.rel_l2(g, ref)
    dT = abs(R.typicality_scalar(g).item() - R.typicality_scalar(ref).item()) / ref.mean().item()
    print(f"fp32 grid on the GPU vs the fp32 CPU oracle @64x64 N=1: rel-L2 {r:.2e}, |dT|/mean loss {dT:.2e}")
    assert g.shape == ref.shape and r < TOL_E2E and dT < 1e-6


def test_fp16_engine_vs_fp32_ground_truth_at_the_baseline_configuration(net32, sd15_weights_f16):
    """BASELINE configs[1] itself — 512 px (64 x 64 latent), N = 10 draws x 2 prompts per image — for four images: the fp16 engine's
    grid and T(x|c) against the exact-fp32 evaluation of the same U-Net on the same inputs (the CPU oracle reaches this size only
    for single forwards).  Bounds: the loss-grid and T(x|c) tolerances of tests/test_gpu_e2e.py (1.5 x the autocast oracle's own
    spread under re-ordering; the fp32 truth is one more sample of that cloud's centre)."""
    from diff_mining_amd.engine import UNetEngine
    from diff_mining_amd.typicality import TypicalityScorer
    import json
    import os
    floor = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_T_floor.json")))
    N, h, w, n_img = 10, 64, 64, 4
    e16 = UNetEngine(0)
    e16.load_state_dict(sd15_weights_f16)
    sc = TypicalityScorer(e16, seed=42, N=N, t_min=0.1, t_max=0.7)
    xs, _, _, c = synth.synth_inputs(n_img, 1, h, w, latent_dtype=np.float32)
    xs, c = torch.from_numpy(xs), torch.from_numpy(c)
    worst = [0.0, 0.0, 0.0]
    for i in range(n_img):
        x = xs[i:i + 1]
        noises, ts = sc.draw(x.shape)
        grid = sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False).float()
        ref = _grid32(net32, x, c, noises, ts)
        rl = U.rel_l2(grid.cpu(), ref.cpu())
        T, T_ref, ml = R.typicality_scalar(grid.cpu()).item(), R.typicality_scalar(ref.cpu()).item(), ref.mean().item()
        d = (rl, abs(T - T_ref) / abs(T_ref), abs(T - T_ref) / ml)
        worst = [max(a, b) for a, b in zip(worst, d)]
        print(f"[64x64 N=10 image {i}] fp16 grid vs fp32 truth rel-L2 {rl:.2e}; T engine {T:.6f} fp32 {T_ref:.6f} mean loss {ml:.4f}: "
              f"|dT|/|T| {d[1]:.2e}  |dT|/mean-loss {d[2]:.2e}")
    e16.close()
    # north_star's own number, asserted: "outputs must match the reference ... to <= 1e-3 rel fp16", with exact fp32
    # as the yardstick — the centre of the cloud every fp16 evaluation of this network (the reference's included) scatters around,
    # DESIGN.md section 2.  Measured r04: 9.00-9.15e-4 for the fp16 grids of these four images; budget rule for new rewrites in
    # DESIGN.md section 2a (a fold ships only if this stays <= 9.5e-4)
    assert worst[0] <= 1.0e-3, f"fp16 loss grid vs exact fp32 at BASELINE configs[1]: {worst[0]:.3e} > 1e-3"
    assert worst[2] <= 1.5 * floor["max_dT_over_mean_loss"] * 1.5     # T10 bound of test_gpu_e2e.py; x 1.5: the grid is fp16-rounded, the truth is not


@pytest.fixture(scope="module")
def vae_sd():
    return synth.synth_vae_state_dict(seed=0, dtype=np.float16)


@pytest.mark.parametrize("B,H,W,draws", [(2, 64, 64, 1), (1, 100, 68, 3), (1, 85, 131, 1)])
def test_vae32_vs_fp32_oracle(net32, vae_sd, B, H, W, draws):
    """`vae.encode(img).latent_dist.sample() * scaling_factor` of the featuriser's fp32 pipeline (dift.py:187,197-199), incl. image
    sizes that are not multiples of 8, against the VAE oracle in fp32 mode."""
    from oracle import vae_ref as V
    if not getattr(net32, "_vae_ready", False):
        net32.load_vae_state_dict(vae_sd)
    sdt = {k: torch.from_numpy(v).float() for k, v in vae_sd.items()}
    img = torch.from_numpy(synth.synth_image(B, H, W)).float()
    noise = _randn(B * draws, 4, H // 8, W // 8, seed=9)
    lat, mom = net32.vae_encode(img, noise, return_moments=True, draws_per_image=draws)
    m_ref = V.vae_moments(sdt, img, autocast=False)
    l_ref = V.posterior_sample(m_ref.repeat_interleave(draws, 0), noise)
    rm, rl = U.rel_l2(mom.cpu(), m_ref), U.rel_l2(lat.cpu(), l_ref)
    mode = net32.vae_encode(img, None).cpu()
    print(f"vae32 {B}x{H}x{W} draws {draws}: moments rel-L2 {rm:.2e}, latents {rl:.2e}")
    assert mom.shape == m_ref.shape and lat.shape == l_ref.shape and rm < TOL_E2E and rl < TOL_E2E
    assert U.rel_l2(mode, V.posterior_sample(m_ref, None)) < TOL_E2E


def test_sdfeaturizer_from_pixels_in_fp32(net32, vae_sd, sd15_weights_torch):
    """SDFeaturizer.forward(img_tensor, prompt_embeds, t=261, up_ft_index=1, ensemble_size) from PIXELS, every stage in the
    reference's fp32: VAE encode -> ensemble of posterior samples -> add_noise -> U-Net tap -> ensemble mean (dift.py:214-232,173-193)."""
    from diff_mining_amd.dift import SDFeaturizer
    from oracle import vae_ref as V
    if not getattr(net32, "_vae_ready", False):
        net32.load_vae_state_dict(vae_sd)
    H = W = 128
    ens = 2
    img = torch.from_numpy(synth.synth_image(1, H, W)).float()
    pe = torch.from_numpy(synth.synth_inputs(1, 1, 8, 8)[3][:1]).float()
    vn, n = _randn(ens, 4, H // 8, W // 8, seed=11), _randn(ens, 4, H // 8, W // 8, seed=12)
    out = SDFeaturizer(net32).forward(img, pe, t=261, up_ft_index=1, ensemble_size=ens, noise=n, vae_noise=vn)
    sdv = {k: torch.from_numpy(v).float() for k, v in vae_sd.items()}
    lat = V.posterior_sample(V.vae_moments(sdv, img, autocast=False).repeat_interleave(ens, 0), vn)
    noisy = R.add_noise(lat, n, torch.tensor(261), R.alphas_cumprod())
    _, mean_ref = R.dift_features(sd15_weights_torch, noisy, 261, pe.expand(ens, -1, -1), 1)
    r = U.rel_l2(out.cpu(), mean_ref)
    print(f"SDFeaturizer from pixels, fp32 end to end: rel-L2 vs the fp32 oracles {r:.2e}")
    assert out.shape == mean_ref.shape and r < TOL_E2E


@pytest.fixture(scope="module")
def clip_sd():
    return synth.synth_clip_state_dict(seed=0, dtype=np.float16)


def test_clip32_matches_transformers_fixture(net32, clip_sd):
    """`pipe.encode_prompt(prompt)[0]` of the featuriser's fp32 pipeline (dift.py:197-199, 222-226) = `CLIPTextModel(input_ids)[0]` in
    fp32: the fp32 net's text tower against the fixture `transformers.CLIPTextModel` ITSELF produced (tests/golden/clip_text.npz — the one
    pinned oracle of the path) and against the CPU restatement in fp32 mode, at fp32 round-off; the fp16 engine's tower is 1.1e-3 from it."""
    import os
    from oracle import clip_ref
    if not getattr(net32, "_clip_ready", False):
        net32.load_clip_state_dict(clip_sd)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_text.npz"))
    ids = torch.from_numpy(g["input_ids"])
    ref = torch.from_numpy(g["last_hidden_state"])
    out = net32.clip_encode(ids)
    assert out.shape == (3, 77, 768) and out.dtype == torch.float32 and out.is_cuda
    r = U.rel_l2(out, ref)
    sdt = {k: torch.from_numpy(v).float() for k, v in clip_sd.items()}
    ro = U.rel_l2(out, clip_ref.clip_text_forward(sdt, ids, autocast=False))
    print(f"fp32 text tower: rel-L2 vs transformers fp32 {r:.2e}, vs the fp32 restatement {ro:.2e}")
    assert r < TOL_E2E and ro < TOL_E2E
    # properties: deterministic; a prompt's states do not depend on its position in the batch; causal (later tokens cannot matter);
    # more prompts than one pass holds (chunked) = the same rows
    ids5 = torch.from_numpy(synth.synth_token_ids(5, seed=11))
    a = net32.clip_encode(ids5)
    assert torch.equal(a, net32.clip_encode(ids5))
    perm = torch.tensor([3, 0, 4, 1, 2])
    assert torch.equal(net32.clip_encode(ids5[perm]), a[perm])
    ids2 = ids5.clone()
    ids2[:, 40:] = 1234
    b = net32.clip_encode(ids2)
    assert torch.equal(a[:, :40], b[:, :40]) and not torch.equal(a[:, 40:], b[:, 40:])
    many = ids5.repeat(27, 1)                                                # 135 prompts > the 128 of one pass
    assert torch.equal(net32.clip_encode(many), a.repeat(27, 1, 1))
    with pytest.raises(ValueError):
        net32.clip_encode(ids5[:, :50])
    with pytest.raises(ValueError):
        net32.clip_encode(ids5, out_dtype=torch.float16)


def test_sdfeaturizer_string_prompt_in_fp32(net32, clip_sd):
    """VERDICT r05 "missing 4": a STRING prompt through an fp32 featuriser ran the text tower on the fp16 engine.  With CLIP weights on the
    fp32 net the whole of `SDFeaturizer.forward(img, prompt: str, ...)` is fp32 (dift.py:214-232); measured here: what the fp16 tower cost —
    the DIFT feature from fp16-tower hidden states against the one from fp32-tower hidden states."""
    from diff_mining_amd.dift import SDFeaturizer
    from diff_mining_amd.engine import UNetEngine
    if not getattr(net32, "_clip_ready", False):
        net32.load_clip_state_dict(clip_sd)
    ids = torch.from_numpy(synth.synth_token_ids(1, seed=5))

    class Tok:                                   # stands in for CLIPTokenizer (the vocabulary is not in the image): fixed ids
        model_max_length = 77
        calls = 0

        def __call__(self, prompts, max_length, padding, truncation, return_tensors):
            assert max_length == 77 and padding == "max_length" and truncation and return_tensors == "pt" and len(prompts) == 1
            Tok.calls += 1
            return type("Enc", (), {"input_ids": ids.long()})()

    f = SDFeaturizer(net32, tokenizer=Tok())
    lat, n = _randn(1, 4, 16, 16, seed=3), _randn(2, 4, 16, 16, seed=4)
    out = f.forward(lat, "A car from the 1970s.", t=261, up_ft_index=1, ensemble_size=2, noise=n)
    ref = SDFeaturizer(net32).forward(lat, net32.clip_encode(ids), t=261, up_ft_index=1, ensemble_size=2, noise=n)
    assert out.dtype == torch.float32 and torch.equal(out, ref) and Tok.calls == 1          # the string took the fp32 tower
    f.forward(lat, "A car from the 1970s.", t=261, up_ft_index=1, ensemble_size=2, noise=n)
    assert Tok.calls == 1                                                                    # cached per distinct string
    e16 = UNetEngine(0)
    try:
        e16.load_clip_state_dict(clip_sd)
        c16 = e16.clip_encode(ids)
    finally:
        e16.close()
    c32 = net32.clip_encode(ids)
    out16 = SDFeaturizer(net32).forward(lat, c16, t=261, up_ft_index=1, ensemble_size=2, noise=n)
    rc, rf = U.rel_l2(c16, c32), U.rel_l2(out16, out)
    cos = F.cosine_similarity(out16[0].flatten(1).T.double(), out[0].flatten(1).T.double(), dim=1).min().item()      # per feature-map cell, over the 1280 channels
    print(f"fp16 text tower inside the fp32 featuriser: hidden states rel-L2 {rc:.2e} -> DIFT feature rel-L2 {rf:.2e}, min per-pixel cosine {cos:.7f}")
    assert 2e-4 < rc < 3e-3 and rf < 3e-3 and cos > 1 - 1e-5


def test_f32_score_conds_takes_the_product_surface_keywords(net32):
    """`TypicalityScorer.compute_losses_batch` / `compute_submission` call `score_conds(..., latent_dtype=, slot_table=)`; the fp32 net
    takes the same keywords: `slot_table` [n_cond, U] = the registered prompt of draw i in its k-th condition (rows k U + i, bit-equal to
    `score` on the tiled batch with those slots), `latent_dtype` None / float32 only."""
    h = w = 8
    x, eps, t, c = synth.synth_inputs(1, 3, h, w, latent_dtype=np.float32)
    x, eps, t, c = torch.from_numpy(x), torch.from_numpy(eps), torch.from_numpy(t), torch.from_numpy(c).float()
    c3 = torch.cat([c, (c[:1] * 0.5 + c[1:2] * 0.5)])                          # three registered prompts
    net32.set_prompts(c3)
    table = torch.tensor([[2, 0, 2], [1, 1, 1]], dtype=torch.int32)
    got = net32.score_conds(x, eps, t, 2, latent_dtype=torch.float32, slot_table=table)
    want = net32.score(x, eps.repeat(2, 1, 1, 1), t.repeat(2), table.reshape(-1))
    assert got.shape == (6, 4, h, w) and torch.equal(got, want)
    plain = net32.score_conds(x, eps, t, 2)
    assert torch.equal(plain, net32.score(x, eps.repeat(2, 1, 1, 1), t.repeat(2), torch.tensor([0, 0, 0, 1, 1, 1], dtype=torch.int32)))
    assert not torch.equal(plain, got)
    with pytest.raises(ValueError, match="latent_dtype"):
        net32.score_conds(x, eps, t, 2, latent_dtype=torch.float16)
    with pytest.raises(ValueError, match="slot_table"):
        net32.score_conds(x, eps, t, 2, slot_table=torch.zeros(5, dtype=torch.int32))


def test_f32_net_error_behaviour(sd15_weights_f16):
    """Failures are loud and named (no fallback): calls before finalize / set_prompts, a state dict with a tensor missing or of the
    wrong shape, a prompt slot outside the registered prompts, a batch that is not whole ensembles."""
    from diff_mining_amd.engine import EngineError, UNetEngineF32
    e = UNetEngineF32(0)
    x = torch.zeros(2, 4, 8, 8)
    with pytest.raises(EngineError, match="finalized"):
        e.n_prompts = 1
        e.unet(x, torch.tensor(5), torch.zeros(2, dtype=torch.int32))
    e.n_prompts = 0
    partial = {k: v for k, v in sd15_weights_f16.items() if k != "mid_block.resnets.1.conv2.weight"}
    with pytest.raises(EngineError, match="mid_block.resnets.1.conv2.weight"):
        e.load_state_dict(partial)
    e.close()
    e = UNetEngineF32(0)
    bad = dict(sd15_weights_f16)
    bad["conv_in.weight"] = bad["conv_in.weight"][:, :3]
    with pytest.raises(EngineError, match="conv_in.weight"):
        e.load_state_dict(bad)
    e.close()
    e = UNetEngineF32(0)
    e.load_state_dict(sd15_weights_f16)
    with pytest.raises(EngineError, match="set_prompts"):
        e.n_prompts = 1                      # get past the host-side slot check: the library itself must refuse
        e.unet(x, torch.tensor(5), torch.zeros(2, dtype=torch.int32))
    e.set_prompts(torch.zeros(2, 77, 768))
    with pytest.raises(EngineError, match="slot"):
        e.unet(x, torch.tensor(5), torch.tensor([0, 2], dtype=torch.int32))
    with pytest.raises(EngineError, match="ensemble"):
        e.dift(torch.zeros(3, 4, 8, 8), torch.tensor(5), torch.zeros(3, dtype=torch.int32), 1, 2)
    with pytest.raises(EngineError, match="VAE"):
        e.vae_encode(torch.zeros(1, 3, 64, 64))
    out = e.unet(x, torch.tensor([5, 900]), torch.tensor([1, 0], dtype=torch.int32))          # and it still works afterwards
    assert out.shape == (2, 4, 8, 8) and torch.isfinite(out).all()
    e.close()


def test_fp32_net_against_real_diffusers_fixture():
    """The pin the image cannot provide: when tests/make_golden_with_diffusers.py has been run where diffusers==0.24.0 exists,
    the fp32 net is compared with diffusers' OWN fp32 U-Net on the CPU — `unet(noisy, t, ctx).sample` at every size of the
    fixture (incl. the odd latents) and the `up_blocks[1]` tap — at fp32 round-off, with the fixture's unrounded fp32 weights."""
    import os
    golden = os.path.join(os.path.dirname(__file__), "golden")
    sp = os.path.join(golden, "score_diffusers.npz")
    if not os.path.exists(sp):
        pytest.skip("tests/golden/score_diffusers.npz absent: run tests/make_golden_with_diffusers.py where diffusers==0.24.0 is installed")
    from diff_mining_amd.engine import UNetEngineF32
    g = np.load(sp)
    net = UNetEngineF32(0)
    net.load_state_dict(synth.synth_state_dict(seed=0, dtype=np.float32))          # the weights the fixture was made with
    for tag in ("8x8", "16x16", "12x10", "32x42", "32x48"):
        if f"pred_fp32_cpu_{tag}" not in g:
            continue
        noisy, t, c = (torch.from_numpy(g[f"{k}_{tag}"]) for k in ("noisy_fp32_cpu", "t", "c"))
        n = noisy.shape[0] // c.shape[0]
        net.set_prompts(c.float())
        pred = net.unet(noisy, torch.cat([t] * c.shape[0]), torch.arange(c.shape[0], dtype=torch.int32).repeat_interleave(n)).cpu()
        r = U.rel_l2(pred, torch.from_numpy(g[f"pred_fp32_cpu_{tag}"]))
        print(f"fp32 net vs diffusers {str(g['diffusers_version'])} [{tag}]: eps_hat rel-L2 {r:.2e}")
        assert r < TOL_E2E, tag
    dp = os.path.join(golden, "dift_diffusers.npz")
    if os.path.exists(dp):
        d = np.load(dp)
        for sfx in ("", "_12x10"):
            if f"feat_f32_full{sfx}" not in d:
                continue
            noisy = torch.from_numpy(d[f"noisy{sfx}"])
            net.set_prompts(torch.from_numpy(d["prompt"]).float())
            feat, _ = net.dift(noisy, torch.tensor(int(d["t"])), torch.zeros(noisy.shape[0], dtype=torch.int32), 1)
            r = U.rel_l2(feat.cpu(), torch.from_numpy(d[f"feat_f32_full{sfx}"]))
            print(f"fp32 net vs diffusers DIFT tap{sfx}: rel-L2 {r:.2e}")
            assert r < TOL_E2E
    net.close()
