"""End-to-end parity of the HIP engine against the CPU oracle on identical (x, t, eps, c) and
identical synthetic SDv1.5 weights, through the C ABI.

Both dtype flows of `SD.compute_loss` are tested: "f32" (the reference's: fp32 latent and draws, fp32 add_noise,
fp32 eps in the MSE — compute.py:91-101,116) and "f16" (fp16 scheduler arithmetic).

Tolerance statement (north_star: "<= 1e-3 rel fp16"): the reference path itself is fp16 autocast, whose own
rounding noise against exact fp32 arithmetic measures ~1.5e-3 rel-L2 on eps_hat with these weights (oracle
autocast-vs-fp32, printed below).  Two independent fp16 implementations therefore cannot agree elementwise to
1e-3; every bound asserted here is <= 2x the value measured on MI355X (r02, printed by each test and tabulated
in DESIGN.md §2), so a regression of 2x fails:
   rel-L2(eps_hat_engine, oracle autocast / fp32), rel-L2(loss grid), and for the score north_star ends on,
   |T_engine - T_oracle| relative to |T| and to the mean loss at N = 10 draws."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diff_mining_amd import synth  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FLOW = {"f32": (np.float32, torch.float32), "f16": (np.float16, torch.float16)}

# Asserted bounds are DERIVED from the oracle's own indeterminacy (not "2 x what an earlier round measured"):
# the fp16-autocast oracle against itself under a pure re-ordering of its fp32 partial sums moves by up to 1.90e-3 (eps_hat) /
# 1.23e-3 (loss) rel-L2 (tests/test_oracle.py::test_oracle_noise_floor_under_summation_order, profiles/r03_oracle_noise_floor.txt);
# the engine is held to 1.5 x that.
TOL_EPS_HAT = 1.5 * 1.90e-3   # eps_hat rel-L2 vs the autocast oracle (engine: 1.80-2.03e-3) and vs the fp32 oracle (1.40-1.61e-3)
TOL_LOSS = 1.5 * 1.23e-3      # loss / grid rel-L2 vs the autocast oracle (engine: 1.14-1.41e-3)
# |dT| / mean loss with 1-4 draws: T is a mean over n_draws * h * w pixel-draws of differences of nearly equal losses, so its noise
# scales with 1 / sqrt(count): the N = 10 @32x32 floor (2.48e-5 over 10 240 pixel-draws) x sqrt(10 240 / 64) for the smallest case
# (one draw @8x8) x 1.5 = 4.7e-4 (engine: <= 2.35e-4)
TOL_T_MEANLOSS = 4.7e-4
# config-1 substitute, 16 images x N = 2 @32x48: the same scaling gives 2.48e-5 x sqrt(10240 / 3072) = 4.5e-5 as the floor's largest
# of six samples; the assertion is on the MAX over 16 images, held to 2.7 x that (engine: 5.9e-5)
TOL_C1_MEANLOSS = 1.2e-4
TOL_DIFT = 1.5 * 1.90e-3 * 1.15   # DIFT feature rel-L2 vs the fp32 oracle: the eps_hat bound; the tap sits at 2/3 of the depth but has no final norm (engine <= 1.62e-3)
# T(x|c) at the BASELINE draw count: the bound is DERIVED, not "2 x what an earlier round measured": 1.5 x the spread of the
# fp16-autocast oracle against ITSELF when only the order of its fp32 partial sums changes (one thread / channels-last / hidden
# channels permuted; the same two images, N = 10 x 2 prompts @32x32; tools/oracle_noise.py floor -> tests/golden/oracle_T_floor.json,
# profiles/r04_oracle_T_noise_floor.txt).  Two fp16 evaluations of the same function cannot be asked to agree better than one of
# them agrees with itself.
with open(os.path.join(GOLDEN, "oracle_T_floor.json")) as _f:
    _T_FLOOR = json.load(_f)
TOL_T10_MEANLOSS = 1.5 * _T_FLOOR["max_dT_over_mean_loss"]   # 1.5 x 2.48e-5 = 3.7e-5 (engine r04: 2.0e-5, 2.0e-5)
TOL_T10_REL = 1.5 * _T_FLOOR["max_dT_over_T"]               # 1.5 x 1.59e-3 = 2.4e-3 (engine r04: 1.20e-3, 1.27e-3)


@pytest.fixture(scope="module")
def engine(sd15_weights_f16):
    from diff_mining_amd.engine import UNetEngine
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    e = UNetEngine(0)
    e.load_state_dict(sd15_weights_f16)
    yield e
    e.close()


def _inputs(h, w, n_draws, n_img=1, flow="f16"):
    x, eps, t, c = synth.synth_inputs(n_img, n_draws, h, w, latent_dtype=FLOW[flow][0])
    return torch.from_numpy(x), torch.from_numpy(eps), torch.from_numpy(t), torch.from_numpy(c)


def _tile(eps, t, c):
    n_cond, N = c.shape[0], eps.shape[0]
    nb = torch.cat([eps] * n_cond)
    tb = torch.cat([t] * n_cond)
    cc = torch.cat([c[k:k + 1].expand(N, -1, -1) for k in range(n_cond)])
    slots = torch.arange(n_cond, dtype=torch.int32).repeat_interleave(N)
    return nb, tb, cc, slots


def _T(loss, N, h, w):
    return R.typicality_scalar(loss.view(2, N, 4, h, w).transpose(0, 1)).item()


@pytest.mark.parametrize("flow", ["f32", "f16"])
@pytest.mark.parametrize("h,w,n_draws", [(8, 8, 2), (16, 16, 2), (12, 10, 1), (32, 32, 1)])
def test_unet_and_loss_vs_oracle(engine, sd15_weights_torch, h, w, n_draws, flow):
    x, eps, t, c = _inputs(h, w, n_draws, flow=flow)
    ldt = FLOW[flow][1]
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    # 1) plain U-Net forward on the same noisy latents (the flow's own add_noise arithmetic, rounded to fp16 once)
    noisy = R.add_noise(x.expand(nb.shape[0], -1, -1, -1), nb, tb).half()
    pred = engine.unet(noisy, tb, slots).float().cpu()
    ref_ac = R.unet_forward(sd15_weights_torch, noisy.float(), tb, cc.float(), autocast=True)
    ref_32 = R.unet_forward(sd15_weights_torch, noisy.float(), tb, cc.float(), autocast=False)
    r_ac, r_32 = U.rel_l2(pred, ref_ac), U.rel_l2(pred, ref_32)
    r_oo = U.rel_l2(ref_ac, ref_32)
    print(f"[{h}x{w} {flow}] eps_hat rel-L2: engine/autocast-oracle {r_ac:.2e}, engine/fp32-oracle {r_32:.2e}, "
          f"autocast-oracle/fp32-oracle {r_oo:.2e}; max|err| {U.max_abs(pred, ref_ac):.2e} of max|ref| {ref_ac.abs().max():.2f}")
    assert not torch.isnan(pred).any()
    assert r_ac < TOL_EPS_HAT and r_32 < TOL_EPS_HAT, (r_ac, r_32)
    # 2) fused score path (add_noise + U-Net + eps-MSE) in this dtype flow
    loss = engine.score(x, nb, tb, slots, latent_dtype=ldt).cpu()
    ref_loss = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=True, latent_dtype=ldt)
    rl = U.rel_l2(loss, ref_loss)
    print(f"[{h}x{w} {flow}] loss rel-L2 {rl:.2e}")
    assert loss.shape == ref_loss.shape and loss.dtype == torch.float32
    assert rl < TOL_LOSS, rl
    # fused path == unfused path bit for bit (same kernels, add_noise fused into conv_in; fp32 eps in the f32 flow)
    loss_unfused = (pred - nb.float()) ** 2
    assert torch.equal(loss, loss_unfused)
    # 3) typicality scalar
    T, T_ref = _T(loss, n_draws, h, w), _T(ref_loss, n_draws, h, w)
    dm = abs(T - T_ref) / ref_loss.mean().item()
    print(f"[{h}x{w} {flow}] T(x|c) engine {T:.5f} oracle {T_ref:.5f} mean loss {ref_loss.mean():.4f}: "
          f"|dT|/|T| {abs(T - T_ref) / abs(T_ref):.2e}, |dT|/mean-loss {dm:.2e}")
    assert dm <= TOL_T_MEANLOSS, dm


def test_dtype_flows_differ_as_the_oracle_says(engine, sd15_weights_torch):
    """The two flows are different computations (fp16-rounded sqrt(1 - acp) differs by up to 7 % at small t): the
    engine's f32-vs-f16 difference must match the oracle's f32-vs-f16 difference at small t, where the coefficient gap
    (sqrt(1 - acp[0]) = 0.029155 vs 0.03125) is largest."""
    x, eps, t, c = _inputs(8, 8, 2, flow="f32")
    t = torch.tensor([0, 5])
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    l32 = engine.score(x, nb, tb, slots, latent_dtype=torch.float32).cpu()
    l16 = engine.score(x.half(), nb.half(), tb, slots, latent_dtype=torch.float16).cpu()
    r32 = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=True, latent_dtype=torch.float32)
    r16 = R.compute_loss(sd15_weights_torch, x.half(), nb.half(), tb, cc, autocast=True, latent_dtype=torch.float16)
    d_eng, d_ref = U.rel_l2(l32, l16), U.rel_l2(r32, r16)
    print(f"t in (0, 5): flow difference engine {d_eng:.2e} oracle {d_ref:.2e}; engine-vs-oracle f32 {U.rel_l2(l32, r32):.2e} "
          f"f16 {U.rel_l2(l16, r16):.2e}")
    # (measured r02: 2.1e-3 both — larger than the engine-vs-oracle distance of either flow, 1.3e-3)
    assert d_eng > 1e-3 and abs(d_eng - d_ref) < 0.25 * d_ref
    assert U.rel_l2(l32, r32) < TOL_LOSS and U.rel_l2(l16, r16) < TOL_LOSS


def test_baseline_shape_vs_oracle(engine, sd15_weights_torch):
    """BASELINE configs[1] geometry (512 px -> 64x64 latent), 1 draw x 2 prompts, against the oracle: the 4096-token
    attention and the 64x64 layers that the small cases do not reach.  (Two samples give <= 256 row tiles, so the
    256x320 igemm tile is NOT reached here: test_gpu_ops.py::test_igemm_conv3x3_big_tile_vs_conv2d and
    test_score_at_baseline_draw_count cover it.)"""
    x, eps, t, c = _inputs(64, 64, 1, flow="f32")
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    loss = engine.score(x, nb, tb, slots).cpu()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=True)
    rl = U.rel_l2(loss, ref)
    T, T_ref = _T(loss, 1, 64, 64), _T(ref, 1, 64, 64)
    dm = abs(T - T_ref) / ref.mean().item()
    print(f"[64x64] loss rel-L2 {rl:.2e}; T(x|c) engine {T:.5f} oracle {T_ref:.5f}; mean loss {ref.mean():.4f}; "
          f"|dT|/|T| {abs(T - T_ref) / abs(T_ref):.2e} |dT|/mean-loss {dm:.2e}")
    assert rl < TOL_LOSS, rl
    assert dm <= TOL_T_MEANLOSS


@pytest.mark.parametrize("n_img,N,h,w", [(2, 10, 32, 32), (1, 4, 64, 64)])
def test_score_at_baseline_draw_count(engine, sd15_weights_torch, n_img, N, h, w):
    """The quantity north_star ends on — T(x|c) = E_N[L_null - L_c] — at the BASELINE draw count (N = 10 draws x 2
    prompts per image; 8 images x N = 10 @64x64 is beyond the CPU oracle's reach, so 2 images x 10 @32x32 and
    1 image x 4 @64x64), through `TypicalityScorer.compute_losses` (dm_score_conds) in the reference's dtype flow.
    Reports |dT|/|T| and |dT|/mean-loss per image."""
    from diff_mining_amd.typicality import TypicalityScorer
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sc = TypicalityScorer(engine, seed=42, N=N, t_min=0.1, t_max=0.7)
    xs, _, _, c = _inputs(h, w, 1, n_img=n_img, flow="f32")
    out = []
    for i in range(n_img):
        x = xs[i:i + 1]
        noises, ts = sc.draw(x.shape)
        assert noises.dtype == torch.float32
        grid = sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False)
        ref = R.compute_losses(sd15_weights_torch, x, c.float(), noises, ts, B=N)
        assert grid.shape == ref.shape == (N, 2, 4, h, w) and grid.dtype == torch.float16
        rl = U.rel_l2(grid.float().cpu(), ref.float())
        T = engine.reduce_typicality(grid)[1].item()
        T_ref = R.typicality_scalar(ref).item()
        ml = ref.float().mean().item()
        out.append((rl, abs(T - T_ref) / abs(T_ref), abs(T - T_ref) / ml))
        print(f"[{h}x{w} N={N} image {i}] grid rel-L2 {rl:.2e}; T engine {T:.6f} oracle {T_ref:.6f} mean loss {ml:.4f}: "
              f"|dT|/|T| {out[-1][1]:.2e}  |dT|/mean-loss {out[-1][2]:.2e}")
        assert rl < TOL_LOSS
        assert out[-1][2] <= (TOL_T10_MEANLOSS if N >= 10 else TOL_T_MEANLOSS)
        if N >= 10:
            assert out[-1][1] <= TOL_T10_REL


def test_config1_substitute_cars_geometry(engine, sd15_weights_torch):
    """BASELINE configs[0] stand-in (the CarDB / CPU-diffusers run itself cannot exist here): 16
    synthetic latents [1,4,32,48] (cars geometry: 256 x 384 px), N = 2 draws, cond vs null -> T(x|c) per image,
    engine vs oracle ("restatement, not diffusers")."""
    from diff_mining_amd.typicality import TypicalityScorer
    torch.set_num_threads(min(16, torch.get_num_threads()))
    n_img, N, h, w = 16, 2, 32, 48
    sc = TypicalityScorer(engine, seed=42, N=N, t_min=0.1, t_max=0.7)
    xs, _, _, c = _inputs(h, w, 1, n_img=n_img, flow="f32")
    noises, ts = sc.draw((1, 4, h, w))
    # all 16 images in one engine call (image-major draws, batched on-device reduction) ...
    engine.set_prompts(c)
    xi = torch.arange(n_img, dtype=torch.int32).repeat_interleave(N)
    loss = engine.score_conds(xs, noises.repeat(n_img, 1, 1, 1), ts.repeat(n_img), 2, x_index=xi)
    maps, T = engine.reduce_typicality_batched(loss, n_img, N, 2, cond_major=True)
    T = T.cpu()
    # ... against the oracle image by image
    T_ref, ml = [], []
    for i in range(n_img):
        ref = R.compute_losses(sd15_weights_torch, xs[i:i + 1], c.float(), noises, ts, B=N)
        T_ref.append(R.typicality_scalar(ref).item())
        ml.append(ref.float().mean().item())
        # same image through the per-image surface: bit-identical to its rows of the batched call
        if i in (0, n_img - 1):
            g = sc.compute_losses(xs[i:i + 1], c, noises=noises, timesteps=ts, to_host=False)
            rows = loss.view(2, n_img, N, 4, h, w)[:, i].transpose(0, 1)
            assert torch.equal(g, rows.half())
            assert torch.allclose(engine.reduce_typicality(rows.contiguous())[0], maps[i], atol=0, rtol=0)
    T_ref, ml = torch.tensor(T_ref), torch.tensor(ml)
    dm = ((T - T_ref).abs() / ml)
    print("config-1 substitute: T engine", [round(v, 5) for v in T.tolist()])
    print("                     T oracle", [round(v, 5) for v in T_ref.tolist()])
    print(f"                     max |dT|/mean-loss {dm.max():.2e}, rank agreement "
          f"{(T.argsort() == T_ref.argsort()).float().mean():.2f}")
    assert dm.max().item() <= TOL_C1_MEANLOSS


def test_golden_grid_fixture(engine):
    """tests/golden/grid_8x8.npz: `D.compute_losses` grid [4,2,4,8,8] from the oracle (CPU-generator draws, seed 42)
    vs the engine, in both dtype flows."""
    from diff_mining_amd.typicality import TypicalityScorer
    g = np.load(os.path.join(GOLDEN, "grid_8x8.npz"))
    x, c = torch.from_numpy(g["x"]), torch.from_numpy(g["c"])
    noises, ts = torch.from_numpy(g["noises"]), torch.from_numpy(g["timesteps"])
    assert x.dtype == torch.float32 and noises.dtype == torch.float32
    sc = TypicalityScorer(engine, seed=42, N=4, t_min=0.1, t_max=0.7)
    n2, t2 = sc.draw(x.shape)
    assert torch.equal(n2, noises) and torch.equal(t2, ts)                 # the scorer draws what the fixture holds
    grid = sc.compute_losses(x, c)
    r32 = U.rel_l2(grid.float(), torch.from_numpy(g["grid"]).float())
    sc16 = TypicalityScorer(engine, seed=42, N=4, t_min=0.1, t_max=0.7, latent_dtype=torch.float16)
    grid16 = sc16.compute_losses(x.half(), c)
    r16 = U.rel_l2(grid16.float(), torch.from_numpy(g["grid_f16flow"]).float())
    print(f"golden grid 8x8 rel-L2: f32 flow {r32:.2e}, f16 flow {r16:.2e}")
    assert grid.shape == (4, 2, 4, 8, 8) and grid.dtype == torch.float16
    assert r32 < TOL_LOSS and r16 < TOL_LOSS


DIFFUSERS_GOLDEN = os.path.join(GOLDEN, "score_diffusers.npz")


@pytest.mark.skipif(not os.path.exists(DIFFUSERS_GOLDEN), reason="tests/golden/score_diffusers.npz absent: generate it "
                    "with tests/make_golden_with_diffusers.py where diffusers 0.24 is installed (it is not in this image)")
def test_against_real_diffusers_fixture(engine):
    """Pins parity to the REAL dependency: inputs + outputs of diffusers' UNet2DConditionModel / scheduler.add_noise /
    F.mse_loss under autocast, produced by tests/make_golden_with_diffusers.py from the same synthetic weights."""
    g = np.load(DIFFUSERS_GOLDEN)
    x, eps, t, c = (torch.from_numpy(g[k]) for k in ("x", "eps", "t", "c"))
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    loss = engine.score(x, nb, tb, slots, latent_dtype=torch.float32).cpu()
    rl = U.rel_l2(loss, torch.from_numpy(g["loss_autocast_cuda"] if "loss_autocast_cuda" in g else g["loss_fp32_cpu"]))
    print(f"vs diffusers {str(g['diffusers_version'])}: loss rel-L2 {rl:.2e}")
    assert rl < TOL_LOSS
    for tag in ("16x16", "12x10", "32x42", "32x48"):              # odd latents (`upsample_size`) and the widest cars latent
        if f"loss_fp32_cpu_{tag}" not in g:
            continue
        x, eps, t, c = (torch.from_numpy(g[f"{k}_{tag}"]) for k in ("x", "eps", "t", "c"))
        nb, tb, cc, slots = _tile(eps, t, c)
        engine.set_prompts(c)
        loss = engine.score(x, nb, tb, slots, latent_dtype=torch.float32).cpu()
        key = f"loss_autocast_cuda_{tag}" if f"loss_autocast_cuda_{tag}" in g else f"loss_fp32_cpu_{tag}"
        rl = U.rel_l2(loss, torch.from_numpy(g[key]))
        print(f"vs diffusers [{tag}]: loss rel-L2 {rl:.2e}")
        assert rl < TOL_LOSS, tag
    dp = os.path.join(GOLDEN, "dift_diffusers.npz")
    if os.path.exists(dp):
        d = np.load(dp)
        for sfx in ("", "_12x10"):
            if f"noisy{sfx}" not in d:
                continue
            noisy = torch.from_numpy(d[f"noisy{sfx}"])
            engine.set_prompts(torch.from_numpy(d["prompt"]))
            feat, _ = engine.dift(noisy.half(), torch.tensor(int(d["t"])), torch.zeros(noisy.shape[0], dtype=torch.int32), 1)
            rf = U.rel_l2(feat.float().cpu(), torch.from_numpy(d[f"feat_fp32{sfx}"]).float())
            print(f"vs diffusers DIFT tap{sfx}: rel-L2 {rf:.2e}")
            assert rf < TOL_DIFT


def test_golden_fixture(engine):
    """Committed golden vectors (tests/golden, produced by tests/make_golden.py from the oracle)."""
    path = os.path.join(GOLDEN, "score_8x8.npz")
    g = np.load(path)
    x, eps, t, c = (torch.from_numpy(g[k]) for k in ("x", "eps", "t", "c"))
    assert x.dtype == torch.float32 and eps.dtype == torch.float32
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    rl32 = U.rel_l2(engine.score(x, nb, tb, slots).cpu(), torch.from_numpy(g["loss_autocast_f32flow"]))
    rl16 = U.rel_l2(engine.score(x.half(), nb.half(), tb, slots).cpu(), torch.from_numpy(g["loss_autocast_f16flow"]))
    print(f"golden 8x8 loss rel-L2: f32 flow {rl32:.2e}, f16 flow {rl16:.2e}")
    assert rl32 < TOL_LOSS and rl16 < TOL_LOSS
    gd = np.load(os.path.join(GOLDEN, "dift_16x16.npz"))
    noisy, tt, pe = torch.from_numpy(gd["noisy"]), int(gd["t"]), torch.from_numpy(gd["prompt"])
    engine.set_prompts(pe)
    feat, mean = engine.dift(noisy, torch.tensor(tt), torch.zeros(noisy.shape[0], dtype=torch.int32), 1, noisy.shape[0])
    ref_ft = torch.from_numpy(gd["feat_fp32"])
    rf = U.rel_l2(feat.float().cpu(), ref_ft)
    rm = U.rel_l2(mean.cpu(), ref_ft.mean(0, keepdim=True))
    print(f"golden dift feat rel-L2 {rf:.2e} mean rel-L2 {rm:.2e}")
    assert feat.shape == ref_ft.shape and rf < TOL_DIFT and rm < TOL_DIFT


def test_dift_vs_oracle(engine, sd15_weights_torch):
    h = w = 16
    ens = 4
    x, eps, t, c = _inputs(h, w, ens)
    acp = R.alphas_cumprod()
    noisy = R.add_noise(x.float().expand(ens, -1, -1, -1), eps.float(), torch.tensor(161), acp).half()
    engine.set_prompts(c[:1])
    slots = torch.zeros(ens, dtype=torch.int32)
    feat, mean = engine.dift(noisy, torch.tensor(161), slots, 1, ens)
    ft_ref, mean_ref = R.dift_features(sd15_weights_torch, noisy.float(), 161, c[:1].float().expand(ens, -1, -1), 1)
    assert feat.shape == (ens, 1280, h // 2, w // 2) and mean.shape == (1, 1280, h // 2, w // 2)
    rf, rm = U.rel_l2(feat.float().cpu(), ft_ref), U.rel_l2(mean.cpu(), mean_ref)
    print(f"dift rel-L2 vs fp32 oracle: features {rf:.2e}, ensemble mean {rm:.2e}")
    assert rf < TOL_DIFT and rm < TOL_DIFT
    # up_ft_index 0 and 2 shapes
    f0, _ = engine.dift(noisy, torch.tensor(161), slots, 0)
    f2, _ = engine.dift(noisy, torch.tensor(161), slots, 2)
    assert f0.shape == (ens, 1280, h // 4, w // 4) and f2.shape == (ens, 640, h, w)


def test_surface_compute_losses_and_reductions(engine, sd15_weights_torch):
    """TypicalityScorer mirrors D.compute_losses: layout [N,2,4,h,w] fp16, cond 0 = c, 1 = null."""
    from diff_mining_amd.typicality import TypicalityScorer
    sc = TypicalityScorer(engine, seed=42, N=3, t_min=0.1, t_max=0.7)
    x, _, _, c = _inputs(8, 8, 1)
    x = x.float()
    noises, ts = sc.draw(x.shape)
    n_ref, t_ref = R.draw_noise_and_timesteps(tuple(x.shape), 3, 0.1, 0.7, seed=42)
    assert noises.dtype == torch.float32 and torch.equal(noises, n_ref) and torch.equal(ts, t_ref)   # same CPU-generator draws
    grid = sc.compute_losses(x, c, noises=noises, timesteps=ts)
    assert grid.shape == (3, 2, 4, 8, 8) and grid.dtype == torch.float16 and grid.device.type == "cpu"
    ref = R.compute_losses(sd15_weights_torch, x, c.float(), noises, ts, B=3)
    assert U.rel_l2(grid.float(), ref.float()) < TOL_LOSS
    # compute_loss with a tiled c tensor (reference calling convention) gives the same numbers
    nb, tb, cc, _ = _tile(noises, ts, c)
    loss = sc.compute_loss(x, nb, tb, cc).cpu()
    g2 = torch.stack(torch.split(loss, [3, 3], dim=0), dim=1).half()
    assert torch.equal(g2, grid)
    # identical prompts in both slots => L[:,0] == L[:,1] bit-exact
    same = sc.compute_losses(x, torch.stack([c[0], c[0]]), noises=noises, timesteps=ts)
    assert torch.equal(same[:, 0], same[:, 1])
    # GPU reductions vs oracle reductions on the same grid
    gpu_grid = grid.to(engine.device)
    hm, scal = engine.reduce_typicality(gpu_grid)
    torch.testing.assert_close(hm.cpu(), R.typicality_map(grid), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(scal.cpu()[0], R.typicality_scalar(grid), atol=1e-5, rtol=1e-4)
    # .npy contract (compute.py:192): <f2, (N,2,4,h,w)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = sc.save_grid(d, "/data/cars/1970__img_001.jpg", grid)
        assert p.endswith("1970__img_001.npy")
        back = np.load(p)
        assert back.dtype == np.float16 and back.shape == (3, 2, 4, 8, 8)


@pytest.mark.parametrize("h,w,H,W,k", [(8, 8, 64, 64, 5), (32, 40, 256, 320, 50), (64, 64, 512, 512, 64), (16, 16, 128, 128, 1)])
def test_image_space_typicality_vs_reference_order(engine, h, w, H, W, k):
    """dm_typicality_image vs the reference's order of operations (cluster.py:125-137) on CPU."""
    g = torch.Generator().manual_seed(h * 1000 + k)
    grid = (torch.rand(3, 2, 4, h, w, generator=g) * 2).half()
    from diff_mining_amd.typicality import TypicalityScorer
    sc = TypicalityScorer(engine)
    got = sc.load_typicality(grid, (H, W), k, k).cpu()
    ref = R.load_typicality(grid, (H, W), k, k)
    assert got.shape == ref.shape == (H - k + 1, W - k + 1)
    torch.testing.assert_close(got, ref, atol=2e-5, rtol=1e-4)
    if k == 1:
        torch.testing.assert_close(sc.pixel_heatmap(grid, (H, W)).cpu(), ref, atol=2e-5, rtol=1e-4)


def test_compute_losses_vs_the_reference_classes_fixture(engine):
    """tests/golden/host_ref.npz: the grid the reference's OWN `D.compute_losses` / `SD.compute_loss` produced (run from /root/reference by
    tests/make_golden_host.py with the fp32 oracle as `pipe.unet`; torch.autocast('cuda') is inert without CUDA, so the flow is fp32):
    N = 7 draws in chunks of B = 3.  The scorer must make the same draws and fill the same [N, 2, 4, h, w] layout; the values agree to the
    fp16-vs-fp32 distance."""
    import json
    from diff_mining_amd.typicality import TypicalityScorer
    a = np.load(os.path.join(GOLDEN, "host_ref.npz"))
    m = json.load(open(os.path.join(GOLDEN, "host_ref.json")))
    sc = TypicalityScorer(engine, seed=m["seed"], N=m["N"], t_min=m["t_min"], t_max=m["t_max"])
    x, c = torch.from_numpy(a["x"]), torch.from_numpy(a["embeds"])
    grid = sc.compute_losses(x, c, B=m["B"])                       # draws its own (eps, t)
    eps, t = sc.draw(x.shape)
    assert np.array_equal(eps.numpy(), a["noises"]) and np.array_equal(t.numpy(), a["timesteps"])
    ref = torch.from_numpy(a["grid"]).float()
    assert grid.shape == ref.shape and grid.dtype == torch.float16
    rl = U.rel_l2(grid.float(), ref)
    print(f"grid vs the reference classes' fp32 grid: rel-L2 {rl:.2e}")
    assert rl < TOL_LOSS
    # layout: every (draw, prompt) cell is closest to its own cell of the reference grid
    for i in range(m["N"]):
        for k in range(2):
            d_own = (grid[i, k].float() - ref[i, k]).norm()
            assert d_own < (grid[i, k].float() - ref[i, 1 - k]).norm() and d_own < (grid[i, k].float() - ref[(i + 1) % m["N"], k]).norm()


def test_consumers_vs_the_reference_fixture(engine):
    """PINNED to the reference's own code: `Cluster.load_typicality`, `load_typicality_norm`, `d_compute`, `normalize` run by
    tests/make_golden_consumers.py from /root/reference (tests/golden/consumers_ref.npz) vs dm_typicality_image +
    dm_normalize_map.  The engine pools the mean map once (the reference pools 2N maps, then subtracts and averages: linear
    operations in another order), so the un-normalised maps agree to summation order; the normalisations are then exact
    functions of those maps."""
    from diff_mining_amd.typicality import TypicalityScorer
    f = np.load(os.path.join(GOLDEN, "consumers_ref.npz"))
    sc = TypicalityScorer(engine)
    for tag in ("a", "b", "c"):
        grid = torch.from_numpy(f[f"{tag}_grid"])
        H, W, k = (int(v) for v in f[f"{tag}_size"])
        got = sc.load_typicality(grid, (H, W), k, k).cpu().numpy()
        ref = f[f"{tag}_load_typicality"]
        scale = np.abs(ref).max()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-6 * max(scale, 1.0), (tag, np.abs(got - ref).max(), scale)
        gn = sc.load_typicality_norm(grid, (H, W)).cpu().numpy()
        rn = f[f"{tag}_load_typicality_norm"]
        # [0, 1] map: |min| and max divide differences of O(0.1), so 2e-6 of the raw map is ~2e-5 here
        assert gn.shape == rn.shape and np.abs(gn - rn).max() <= 5e-5, (tag, np.abs(gn - rn).max())
        box = [int(v) for v in f[f"{tag}_box"]]
        gd = sc.d_compute(grid, H, W, *box).cpu().numpy()
        rd = f[f"{tag}_d_compute"]
        assert gd.shape == rd.shape and np.abs(gd - rd).max() <= 5e-5, (tag, np.abs(gd - rd).max())
        print(f"consumers [{tag}] max |d|: map {np.abs(got - ref).max():.2e} norm {np.abs(gn - rn).max():.2e} d_compute {np.abs(gd - rd).max():.2e}")
    # the normalisations themselves, on the reference's own map: bit-exact (fp32 IEEE division, order-free min / max)
    dm = torch.from_numpy(f["a_load_typicality_k1"])
    assert np.array_equal(engine.normalize_map(dm, "signed").cpu().numpy(), f["a_load_typicality_norm"])
    assert np.array_equal(engine.normalize_map(dm, "positive").cpu().numpy(), f["a_cnorm_positive"])
    assert np.array_equal(engine.normalize_map(dm, "maxabs").cpu().numpy(), f["a_unorm"])
    pos, neg = engine.normalize_map(dm, "split")
    assert np.array_equal(pos.cpu().numpy(), f["a_cnorm_split_pos"]) and np.array_equal(neg.cpu().numpy(), f["a_cnorm_split_neg"])
    # rank_images' scalar (cluster.py:517-531): mean of the image-size per-pixel map
    for tag in ("a", "b", "c"):
        grid = torch.from_numpy(f[f"{tag}_grid"])
        H, W, _ = (int(v) for v in f[f"{tag}_size"])
        hm = sc.pixel_heatmap(grid, (H, W))
        assert abs(hm.mean().item() - float(f[f"{tag}_rank_score"])) <= 2e-6
        # the X-ray application's `dm_pixel` (applications/xray/compute.py:210-218), from the reference's own method
        assert np.abs(hm.cpu().numpy() - f[f"{tag}_xray_dm_pixel"]).max() <= 2e-6


def test_safetensors_checkpoint_round_trip(engine, sd15_weights_f16, tmp_path):
    """`unet/diffusion_pytorch_model.safetensors` (what the reference's --export-only writes,
    finetuning/base.py:245-250) loads into a second engine and scores bit-identically."""
    from safetensors.numpy import save_file
    from diff_mining_amd.engine import UNetEngine, EngineError
    path = str(tmp_path / "diffusion_pytorch_model.safetensors")
    save_file(sd15_weights_f16, path)
    e2 = UNetEngine(0)
    e2.load_safetensors(path)
    x, eps, t, c = _inputs(8, 8, 1)
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    e2.set_prompts(c)
    assert torch.equal(engine.score(x, nb, tb, slots), e2.score(x, nb, tb, slots))
    e2.close()
    # a checkpoint with a missing / misshapen tensor is rejected with a named error
    bad = dict(sd15_weights_f16)
    bad.pop("mid_block.resnets.0.conv1.weight")
    e3 = UNetEngine(0)
    with pytest.raises(EngineError, match="mid_block.resnets.0.conv1.weight"):
        e3.load_state_dict(bad)
    e3.close()
    bad = dict(sd15_weights_f16)
    bad["conv_in.weight"] = bad["conv_in.weight"][:, :3]
    e4 = UNetEngine(0)
    with pytest.raises(EngineError, match="conv_in.weight"):
        e4.load_state_dict(bad)
    e4.close()


def test_unet_callable_drop_in(engine, sd15_weights_torch):
    from diff_mining_amd.typicality import UNetCallable
    unet = UNetCallable(engine)
    x, eps, t, c = _inputs(8, 8, 2)
    nb, tb, cc, slots = _tile(eps, t, c)
    out = unet(nb, tb, cc).sample
    engine.set_prompts(c)
    direct = engine.unet(nb, tb, slots)
    assert out.shape == (4, 4, 8, 8) and out.dtype == torch.float16
    assert torch.equal(out, direct)


def test_determinism_and_batch_invariance(engine):
    """Scored outputs are run-to-run bit-identical (no float atomics) and do not depend on where
    in the batch a sample sits."""
    x, eps, t, c = _inputs(16, 16, 3)
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    a = engine.score(x, nb, tb, slots)
    b = engine.score(x, nb, tb, slots)
    assert torch.equal(a, b)
    perm = torch.tensor([4, 2, 0, 5, 1, 3])
    p = engine.score(x, nb[perm], tb[perm], slots[perm])
    assert torch.equal(p, a[perm.to(a.device)])


@pytest.mark.parametrize("h,w,U,n_img", [(8, 8, 3, 1), (16, 16, 6, 2), (12, 10, 2, 1)])
def test_shared_draw_scoring_is_bit_identical(engine, h, w, U, n_img):
    """dm_score_conds (prompt-independent head of the U-Net evaluated once per draw) == dm_score on
    the cond-major tiled batch, bit for bit."""
    x, eps, t, c = _inputs(h, w, U, n_img)
    engine.set_prompts(c)
    xi = (torch.arange(U) % n_img).int()
    nb, tb, cc, slots = _tile(eps, t, c)
    ref = engine.score(x, nb, tb, slots, x_index=torch.cat([xi, xi]))
    got = engine.score_conds(x, eps, t, 2, x_index=xi)
    assert got.shape == ref.shape
    assert torch.equal(got, ref)
    # three prompts
    c3 = torch.cat([c, (c[:1] * 0.5)])
    engine.set_prompts(c3)
    nb3, tb3 = torch.cat([eps] * 3), torch.cat([t] * 3)
    slots3 = torch.arange(3, dtype=torch.int32).repeat_interleave(U)
    ref3 = engine.score(x, nb3, tb3, slots3, x_index=torch.cat([xi] * 3))
    assert torch.equal(engine.score_conds(x, eps, t, 3, x_index=xi), ref3)


@pytest.mark.parametrize("h,w,n,N", [(16, 16, 5, 3), (64, 64, 9, 10)])
def test_compute_losses_batch_is_bit_equal_to_the_per_image_calls(engine, h, w, n, N):
    """the benchmarked path is the product surface.  `TypicalityScorer.compute_losses_batch` scores n images,
    each under ITS OWN category prompt and the shared null prompt (the reference's work list, compute.py:284-290 -> D.compute
    :182-192), in one engine call (dm_score_conds_slots) -> [n, N, 2, 4, h, w] fp16; every image's slice must be BIT-equal to
    its own `compute_losses` call (a sample's bits do not depend on the batch it rides in, nor on the slot its prompt sits in).
    (64, 64, 9, 10): 90 draws x 2 prompts = two engine chunks (80 + 10 draws): the chunk-local slot tables."""
    from diff_mining_amd.typicality import TypicalityScorer
    x, _, _, c = _inputs(h, w, 1, n_img=n, flow="f32")
    g = torch.Generator().manual_seed(123)
    cats = torch.randn(n, 77, 768, generator=g).half()
    cats[3] = cats[1]                                               # two images of one category
    emb = torch.stack([torch.stack([cats[j], c[1]]) for j in range(n)])       # [n, 2, 77, 768]
    sc = TypicalityScorer(engine, seed=42, N=N, t_min=0.1, t_max=0.7)
    d = U.dev()
    grids = sc.compute_losses_batch(x.to(d), emb.to(d), to_host=False)
    assert grids.shape == (n, N, 2, 4, h, w) and grids.dtype == torch.float16
    assert engine.n_prompts == n                                    # n - 1 distinct categories + the null prompt
    again = sc.compute_losses_batch(x.to(d), emb.to(d), to_host=False)
    assert torch.equal(grids, again)
    for j in range(n):
        want = sc.compute_losses(x[j:j + 1].to(d), emb[j].to(d), to_host=False)
        assert torch.equal(grids[j], want), f"image {j}: batched grid differs from its own compute_losses call"
    # the same images under ONE prompt set (no slot table) and the scalars of the batched reduction
    g1 = sc.compute_losses_batch(x.to(d), emb[0].to(d), to_host=False)
    assert torch.equal(g1[0], grids[0]) and torch.equal(g1[2], sc.compute_losses(x[2:3].to(d), emb[0].to(d), to_host=False))
    _, sc_b = engine.reduce_typicality_batched(grids, n, N, 2)
    for j in range(n):
        assert abs(sc_b[j].item() - sc.typicality_scalar(grids[j]).item()) <= 1e-6 * max(1.0, abs(sc_b[j].item()))
    if h <= 16:
        # a single condition per image (n_cond = 1: dm_score with the slot of each image's prompt) and per-image draws
        g_one = sc.compute_losses_batch(x.to(d), emb[:, :1].to(d), to_host=False)
        assert g_one.shape == (n, N, 1, 4, h, w)
        for j in (0, 3, n - 1):
            assert torch.equal(g_one[j], sc.compute_losses(x[j:j + 1].to(d), emb[j, :1].to(d), to_host=False))
            assert torch.equal(g_one[j, :, 0], grids[j, :, 0])          # = the category column of the two-condition grid
        gen = torch.Generator().manual_seed(5)
        noises = torch.randn(n, 2, 4, h, w, generator=gen)
        ts = torch.randint(100, 700, (n, 2), generator=gen)
        g_pd = sc.compute_losses_batch(x.to(d), emb.to(d), noises=noises, timesteps=ts, to_host=False)
        for j in (1, n - 1):
            assert torch.equal(g_pd[j], sc.compute_losses(x[j:j + 1].to(d), emb[j].to(d), noises=noises[j], timesteps=ts[j], to_host=False))


def test_full_size_properties(engine):
    """BASELINE config 2 shape (64x64 latent, 10 t x 2 prompts): size-independent properties."""
    x, eps, t, c = _inputs(64, 64, 10)
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(torch.stack([c[0], c[0]]))
    loss = engine.score(x, nb, tb, slots)
    assert loss.shape == (20, 4, 64, 64)
    assert torch.isfinite(loss).all()
    assert torch.equal(loss[:10], loss[10:])              # same prompt in both slots
    engine.set_prompts(c)
    loss2 = engine.score(x, nb, tb, slots)
    assert not torch.equal(loss2[:10], loss2[10:])        # conditioning matters
    assert torch.equal(loss2[:10], loss[:10])             # slot 0 unchanged by what slot 1 holds
    assert torch.equal(engine.score_conds(x, eps, t, 2), loss2)     # shared-draw path at the full shape
    m = loss2.mean().item()
    assert 0.1 < m < 10.0, m
    json.dump({"mean_loss_64x64": m}, open(os.path.join(os.environ.get("GRAFT_OUT", "/tmp"), "full_size.json"), "w"))


def test_dift_patch_embeddings(engine):
    """cluster.py:291-299 on the GPU: all patches of an image in one launch vs the numpy restatement;
    the per-image cache computes the map once."""
    from diff_mining_amd.dift import SDFeaturizer, feature_boxes
    from oracle import unet_ref as R
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(1, 1280, 32, 24, generator=g)
    image_hw = (512, 384)
    boxes = [(0, 0, 512, 384), (100, 50, 300, 250), (17, 33, 81, 97), (448, 320, 512, 384), (5, 5, 37, 21)]
    fz = SDFeaturizer(engine)
    out = fz.patch_embeddings(feat, boxes, image_hw).cpu().numpy()
    assert out.shape == (5, 1280)
    for i, b in enumerate(boxes):
        ref = R.dift_patch_embedding(feat[0].numpy().astype(np.float64), b, image_hw)
        assert np.abs(out[i] - ref).max() < 2e-6, (i, np.abs(out[i] - ref).max())
        assert abs(np.linalg.norm(out[i]) - 1) < 1e-5
    assert feature_boxes([(100, 50, 300, 250)], image_hw, (32, 24)).tolist() == [[6, 18, 3, 15]]
    calls = []

    def compute():
        calls.append(1)
        return feat
    a = fz.patch_embeddings(("img0", "prompt", 261), boxes[:2], image_hw, compute)
    b = fz.patch_embeddings(("img0", "prompt", 261), boxes[2:], image_hw, compute)
    assert len(calls) == 1 and torch.equal(torch.cat([a, b]).cpu(), torch.from_numpy(out))
    # empty window -> NaN like numpy's mean of an empty slice
    assert torch.isnan(fz.patch_embeddings(feat, [(10, 10, 10, 40)], image_hw)).all()


def test_chunked_calls_are_batch_independent(engine):
    """More work than one U-Net batch holds (workspace chunking of dm_score / dm_score_conds, and the
    split-K layers): 12 draws x 2 prompts at 128x128 latents (chunks of 40 / 20 samples) equal the same
    samples scored in other groupings, bit for bit."""
    x, eps, t, c = _inputs(128, 128, 12)
    engine.set_prompts(c)
    nb, tb, cc, slots = _tile(eps, t, c)
    full = engine.score(x, nb, tb, slots)                                  # 24 samples -> one chunk of 24
    shared = engine.score_conds(x, eps, t, 2)                              # 12 draws -> one shared chunk
    assert torch.equal(shared, full)
    part = torch.cat([engine.score(x, nb[:7], tb[:7], slots[:7]), engine.score(x, nb[7:], tb[7:], slots[7:])])
    assert torch.equal(part, full)
    # 64x64, 100 draws x 2 prompts = 200 samples: 160 + 40 in dm_score, 80 + 20 draws in dm_score_conds
    x, eps, t, c = _inputs(64, 64, 100)
    nb, tb, cc, slots = _tile(eps, t, c)
    a = engine.score_conds(x, eps, t, 2)
    b = engine.score(x, nb, tb, slots)
    assert a.shape == (200, 4, 64, 64) and torch.equal(a, b)


def test_dift_descriptor_deviation_vs_fp32_oracle(engine, sd15_weights_torch):
    """The reference's DIFT U-Net is fp32 (dift.py:197-199: no torch_dtype, no autocast); the engine is fp16 storage /
    fp32 accumulate.  What that costs where it matters: the L2-normalised patch descriptors `cluster.py:291-299` feeds
    KMeans, at 32x32 latents (256 px images), ensemble 2 — cosine between engine and fp32-oracle descriptors."""
    from diff_mining_amd.dift import SDFeaturizer
    torch.set_num_threads(min(16, torch.get_num_threads()))
    h = w = 32
    ens = 2
    x, eps, _, c = _inputs(h, w, ens, flow="f32")
    fz = SDFeaturizer(engine)
    mean = fz.forward(x, c[:1], t=261, up_ft_index=1, ensemble_size=ens, noise=eps)          # [1,1280,16,16] fp32
    noisy = R.add_noise(x.expand(ens, -1, -1, -1), eps, torch.tensor(261))                   # fp32, as the reference
    _, mean_ref = R.dift_features(sd15_weights_torch, noisy, 261, c[:1].float().expand(ens, -1, -1), 1, autocast=False)
    assert mean.shape == mean_ref.shape == (1, 1280, h // 2, w // 2)
    image_hw = (h * 8, w * 8)
    boxes = [(r, cc_, r + 64, cc_ + 64) for r in range(0, 256, 64) for cc_ in range(0, 256, 64)]       # 16 patches of 64 px
    boxes += [(0, 0, 256, 256), (96, 96, 160, 160), (16, 48, 80, 112)]
    d_eng = fz.patch_embeddings(mean, boxes, image_hw).cpu().double()
    d_ref = torch.stack([torch.from_numpy(R.dift_patch_embedding(mean_ref[0].numpy().astype(np.float64), b, image_hw))
                         for b in boxes])
    cos = (d_eng * d_ref).sum(1)
    rf = U.rel_l2(mean.cpu(), mean_ref)
    print(f"DIFT fp16 engine vs fp32 oracle @32x32: feature-map rel-L2 {rf:.2e}; descriptor cosine min {cos.min():.7f} "
          f"mean {cos.mean():.7f}; max |d_eng - d_ref| {(d_eng - d_ref).abs().max():.2e}")
    assert rf < TOL_DIFT                                   # measured 1.21e-3
    assert cos.min().item() > 1 - 1e-6                     # measured 1 - 5e-7
    json.dump({"dift_descriptor_cos_min": cos.min().item(), "dift_feature_rel_l2": rf},
              open(os.path.join(os.environ.get("GRAFT_OUT", "/tmp"), "dift_dev.json"), "w"))


def test_dift_full_size_properties(engine):
    """BASELINE configs[3] at full size — 8 latents x ensemble 8 = batch 64 @64x64, tap up_blocks[1] -> [64,1280,32,32]:
    run-to-run determinism, batch-position invariance, ensemble mean == mean of the per-sample features."""
    n_lat, ens, lat = 8, 8, 64
    x, eps, _, c = _inputs(lat, lat, ens, n_img=n_lat, flow="f32")
    a = float(R.alphas_cumprod()[161])
    noisy = ((a ** 0.5) * x.repeat_interleave(ens, 0) + ((1 - a) ** 0.5) * eps.repeat(n_lat, 1, 1, 1)).half()
    engine.set_prompts(c[:1])
    slots = torch.zeros(n_lat * ens, dtype=torch.int32)
    tt = torch.tensor(161)
    feat, mean = engine.dift(noisy, tt, slots, 1, ens)
    assert feat.shape == (64, 1280, 32, 32) and feat.dtype == torch.float16
    assert mean.shape == (8, 1280, 32, 32) and mean.dtype == torch.float32
    assert torch.isfinite(feat.float()).all() and torch.isfinite(mean).all()
    feat2, mean2 = engine.dift(noisy, tt, slots, 1, ens)
    assert torch.equal(feat, feat2) and torch.equal(mean, mean2)                       # deterministic
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(5))
    featp, _ = engine.dift(noisy[perm], tt, slots, 1)
    assert torch.equal(featp, feat[perm.to(feat.device)])                              # batch-position invariant
    ref_mean = feat.float().view(n_lat, ens, 1280, 32, 32).mean(1)
    torch.testing.assert_close(mean, ref_mean, atol=1e-6, rtol=1e-6)                   # ensemble mean of dift.py:231
    one, _ = engine.dift(noisy[8:16], tt, slots[:8], 1)                                # a sample's features do not
    assert torch.equal(one, feat[8:16])                                                # depend on the batch it rides in


def test_two_engines_on_one_gpu_two_streams(engine, sd15_weights_f16):
    """What a one-GPU box can check of the multi-engine contract (the two-GPU test below is skipped there): two engines on the SAME
    device, each on a stream of its own, their U-Net runs enqueued back to back so the persistent kernels of both are in flight
    together — per-engine tile hand-out counters, per-engine workspace and K/V cache, the launcher-side zero page.  Both must give
    the bits of a lone run, repeatedly."""
    from diff_mining_amd.engine import UNetEngine
    x, eps, t, c = _inputs(32, 32, 3, flow="f32")
    nb, tb, cc, slots = _tile(eps, t, c)
    dev = engine.device
    xd, nbd, tbd, sld = x.to(dev), nb.to(dev), tb.to(dev), slots.to(dev)
    engine.set_prompts(c)
    ref = engine.score(xd, nbd, tbd, sld, latent_dtype=torch.float32).clone()
    e2 = UNetEngine(0)
    e2.load_state_dict(sd15_weights_f16)
    e2.set_prompts(c)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            a = engine.score(xd, nbd, tbd, sld, latent_dtype=torch.float32)
        with torch.cuda.stream(s2):
            b = e2.score(xd, nbd, tbd, sld, latent_dtype=torch.float32)
        outs.append((a, b))
    torch.cuda.synchronize()
    for a, b in outs:
        assert torch.equal(a, ref) and torch.equal(b, ref)
    e2.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_two_engines_in_one_process(sd15_weights_f16):
    """Kernel function attributes (dynamic LDS size) are per device: a second engine on another GPU of the same
    process must set them again (r01 used a process-global flag)."""
    from diff_mining_amd.engine import UNetEngine
    x, eps, t, c = _inputs(16, 16, 2)
    nb, tb, cc, slots = _tile(eps, t, c)
    outs = []
    for d in (0, 1):
        e = UNetEngine(d)
        e.load_state_dict(sd15_weights_f16)
        e.set_prompts(c)
        with torch.cuda.device(d):
            outs.append(e.score(x, nb, tb, slots).cpu())
        e.close()
    assert torch.equal(outs[0], outs[1])


# ---- r03: the parity holes VERDICT r02 names (next #1a, #1d) -----------------------------------------------------------
def test_xray_config4_vs_oracle(engine, sd15_weights_torch):
    """BASELINE configs[4] (X-ray: 1024 px -> latent 128 x 128, 16 384-token self-attention) against the oracle: 2 draws x 2
    prompts through dm_score in the reference's dtype flow, t from the X-ray range [0, 1000) (applications/xray/compute.py:
    103) with one t < 10 (where sqrt(1 - acp) is smallest and the fp32-vs-fp16 scheduler arithmetic differs most), and the
    per-pixel heat-map of `Typicallity.compute` (xray/compute.py:210-218: mean over latent C, bilinear to the image,
    L_null - L_c, mean over N) against the oracle's reduction of the oracle's grid."""
    h = w = 128
    x, eps, _, c = _inputs(h, w, 2, flow="f32")
    t = torch.tensor([3, 742])
    nb, tb, cc, slots = _tile(eps, t, c)
    engine.set_prompts(c)
    loss = engine.score(x, nb, tb, slots, latent_dtype=torch.float32).cpu()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=True, latent_dtype=torch.float32)
    rl = U.rel_l2(loss, ref)
    per_t = [U.rel_l2(loss[[i, 2 + i]], ref[[i, 2 + i]]) for i in range(2)]
    T, T_ref = _T(loss, 2, h, w), _T(ref, 2, h, w)
    dm = abs(T - T_ref) / ref.mean().item()
    print(f"[128x128 f32] loss rel-L2 {rl:.2e} (t=3: {per_t[0]:.2e}, t=742: {per_t[1]:.2e}); T(x|c) engine {T:.6f} oracle {T_ref:.6f} "
          f"mean loss {ref.mean():.4f}: |dT|/mean-loss {dm:.2e}")
    assert torch.isfinite(loss).all() and rl < TOL_LOSS and max(per_t) < TOL_LOSS, (rl, per_t)
    assert dm <= TOL_T_MEANLOSS, dm
    # heat-maps: latent-grid map and the 1024 x 1024 per-pixel map, engine grid -> engine reduction vs oracle grid -> oracle reduction
    from diff_mining_amd.typicality import TypicalityScorer
    sc = TypicalityScorer(engine)
    grid = loss.view(2, 2, 4, h, w).transpose(0, 1).contiguous()               # [N, n_cond, 4, h, w], fp32 like dm_score's output
    grid_ref = ref.view(2, 2, 4, h, w).transpose(0, 1).contiguous()
    hm = sc.heatmap(grid.to(engine.device)).cpu()
    hm_ref = R.typicality_map(grid_ref)
    px = sc.pixel_heatmap(grid.half().to(engine.device), (1024, 1024)).cpu()    # the reference reduces the stored fp16 grid
    px_ref = R.load_typicality(grid_ref.half(), (1024, 1024), 1, 1)
    # a heat-map pixel is a difference of two nearly equal losses: its error is stated against the loss scale it is made of
    scale = ref.mean().item()
    e_hm, e_px = (hm - hm_ref).abs().max().item() / scale, (px - px_ref).abs().max().item() / scale
    c_hm = torch.corrcoef(torch.stack([hm.flatten(), hm_ref.flatten()]))[0, 1].item()
    print(f"[128x128] heat-map max |d| / mean loss: latent grid {e_hm:.2e}, 1024 px {e_px:.2e}; correlation with the oracle's map {c_hm:.5f}; "
          f"map std / mean loss {hm_ref.std().item() / scale:.2e}")
    assert px.shape == (1024, 1024) and hm.shape == (h, w)
    # measured r03: 5.05e-3 / 5.02e-3, correlation 0.99997 (the map's own std is 0.136 of the mean loss)
    assert e_hm < 1e-2 and e_px < 1e-2 and c_hm > 0.9999, (e_hm, e_px, c_hm)


def test_single_condition_grid_vs_oracle(engine, sd15_weights_torch):
    """The n_cond == 1 branch of `compute_losses` (typicality.py; D.compute_losses with a single embedding row): dm_score on
    slot 0 instead of the shared-draw schedule."""
    from diff_mining_amd.typicality import TypicalityScorer
    sc = TypicalityScorer(engine, seed=42, N=3, t_min=0.1, t_max=0.7)
    x, _, _, c = _inputs(16, 16, 1, flow="f32")
    noises, ts = sc.draw(x.shape)
    grid = sc.compute_losses(x, c[:1], noises=noises, timesteps=ts)
    assert grid.shape == (3, 1, 4, 16, 16) and grid.dtype == torch.float16
    ref = R.compute_losses(sd15_weights_torch, x, c[:1].float(), noises, ts, B=3)
    r = U.rel_l2(grid.float(), ref.float())
    print(f"n_cond = 1 grid rel-L2 {r:.2e}")
    assert r < TOL_LOSS, r
    # and it is the first column of the two-condition grid, bit for bit (a sample's loss does not depend on its batch)
    both = sc.compute_losses(x, c, noises=noises, timesteps=ts)
    assert torch.equal(both[:, :1], grid)


@pytest.mark.parametrize("idx,shape", [(0, (1280, 4, 4)), (2, (640, 16, 16)), (3, (320, 16, 16))])
def test_dift_other_taps_vs_oracle(engine, sd15_weights_torch, idx, shape):
    """`up_ft_indices` other than the paper's 1 (dift.py:133-165): values, not only shapes, against the fp32 oracle
    (up_blocks[0] and [2] include their upsampler, up_blocks[3] has none)."""
    h = w = 16
    ens = 2
    x, eps, t, c = _inputs(h, w, ens)
    noisy = R.add_noise(x.float().expand(ens, -1, -1, -1), eps.float(), torch.tensor(161), R.alphas_cumprod()).half()
    engine.set_prompts(c[:1])
    slots = torch.zeros(ens, dtype=torch.int32)
    feat, mean = engine.dift(noisy, torch.tensor(161), slots, idx, ens)
    ft_ref, mean_ref = R.dift_features(sd15_weights_torch, noisy.float(), 161, c[:1].float().expand(ens, -1, -1), idx)
    assert tuple(feat.shape[1:]) == shape == tuple(ft_ref.shape[1:])
    rf, rm = U.rel_l2(feat.float().cpu(), ft_ref), U.rel_l2(mean.cpu(), mean_ref)
    print(f"dift up_ft_index {idx}: features rel-L2 {rf:.2e}, ensemble mean {rm:.2e}")
    assert rf < TOL_DIFT and rm < TOL_DIFT


def test_engine_loads_from_the_shared_weight_slab(engine, sd15_weights_f16, tmp_path):
    """bench.py --gpus N (VERDICT r05 #8b): rank 0 writes the synthetic state dict once as one slab, the other ranks MAP it and hand the
    engine read-only views into the file.  The full 1.7 GB dict through the real loader: an engine loaded from the mapped views scores
    bit-equal to the engine loaded from the arrays, and the file is gone afterwards."""
    from diff_mining_amd.engine import UNetEngine
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    path = os.path.join(shm, f"dm_test_slab_{os.getpid()}.slab")
    try:
        synth.save_slab(sd15_weights_f16, path)
        mapped = synth.load_slab(path)
        assert list(mapped) == list(sd15_weights_f16) and not mapped["conv_in.weight"].flags.writeable
        assert os.path.getsize(path) >= sum(v.nbytes for v in sd15_weights_f16.values())
        e2 = UNetEngine(0)
        e2.load_state_dict(mapped)
    finally:
        synth.remove_slab(path)
    assert not os.path.exists(path) and not os.path.exists(path + ".json")
    x, eps, t, c = _inputs(16, 16, 2, flow="f32")
    try:
        outs = []
        for e in (engine, e2):
            e.set_prompts(c)
            outs.append(e.score_conds(x, eps, t, 2, latent_dtype=torch.float32).cpu())
        assert torch.equal(outs[0], outs[1])
    finally:
        e2.close()


def test_no_allocation_in_steady_state(engine):
    """no allocation (and no host walk of the schedule) on the steady-state path.  A cars-like stream — latents of
    32 x 40 ... 32 x 48 (256 px short side, varying width), 4 draws x 2 prompts each, prompt sets of varying size — after
    dm_engine_reserve for the largest call: zero device allocations, and one schedule dry run per distinct shape only."""
    from diff_mining_amd.typicality import TypicalityScorer
    widths = [40, 42, 48, 45, 40, 48, 42, 44]
    engine.reserve(max_batch=8, h=32, w=48, n_cond=2, max_prompts=12)
    g = torch.Generator().manual_seed(1)
    c_small, c_big = torch.randn(2, 77, 768, generator=g).half(), torch.randn(9, 77, 768, generator=g).half()
    sc = TypicalityScorer(engine, seed=42, N=4, t_min=0.1, t_max=0.7)
    for wd in sorted(set(widths)):                                   # warm-up: every shape once (its dry run)
        sc.compute_losses(torch.randn(1, 4, 32, wd, generator=g), c_small, to_host=False)
    torch.cuda.synchronize()
    s0 = engine.stats()
    outs = []
    for i, wd in enumerate(widths * 2):
        engine.set_prompts(c_big if i % 3 == 0 else c_small)        # a larger prompt set now and then
        outs.append(sc.compute_losses(torch.randn(1, 4, 32, wd, generator=g), c_small, to_host=False))
    torch.cuda.synchronize()
    s1 = engine.stats()
    print(f"steady state over {2 * len(widths)} images of {len(set(widths))} shapes: device allocations {s1['device_allocs'] - s0['device_allocs']}, "
          f"schedule dry runs {s1['schedule_dry_runs'] - s0['schedule_dry_runs']} (before: {s0})")
    assert s1["device_allocs"] == s0["device_allocs"]
    assert s1["schedule_dry_runs"] == s0["schedule_dry_runs"]
    assert all(torch.isfinite(o.float()).all() for o in outs)


def test_hipgraph_replay_is_bit_identical(engine):
    """Option "graph" (SURVEY §7 step 7): a U-Net run whose schedule key and pointer arguments repeat is captured as a hipGraph
    on its second occurrence and replayed afterwards — the same kernels with the same arguments, so the same bits; prompts may
    change between replays (the K/V cache is read, not captured)."""
    lib = engine.lib
    x, eps, t, c = _inputs(16, 16, 3, flow="f32")
    dev = engine.device
    xd, ed, td = x.to(dev), eps.to(dev), t.to(dev)
    engine.set_prompts(c)
    want = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
    c2 = torch.flip(c, dims=[0]).contiguous()
    engine.set_prompts(c2)
    want2 = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
    assert not torch.equal(want, want2)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g0 = engine.stats()["graph_launches"]
    try:
        assert lib.dm_set_option(b"graph", 1) == 0
        with torch.cuda.stream(side):
            engine.set_prompts(c)
            outs = []
            for i in range(4):
                got = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32)
                side.synchronize()
                outs.append(got.clone())
                del got                                   # the caching allocator hands the same block to the next call
            engine.set_prompts(c2)
            got2 = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
            side.synchronize()
    finally:
        lib.dm_set_option(b"graph", 0)
    launches = engine.stats()["graph_launches"] - g0
    print(f"graph launches {launches} of 5 calls")
    assert launches >= 2, launches
    assert all(torch.equal(o, want) for o in outs)
    assert torch.equal(got2, want2)


def test_tap_reuse_option_end_to_end(engine):
    """Option "tap_reuse" end to end at a 64x64 latent (the geometry whose 64- and 32-pixel-wide convolutions take the tap-reuse
    tile): the loss grids with the option off / on (default rule) / on for every eligible layer agree to fp32 summation order,
    a sample's bits do not depend on the batch it rides in with the option on, and the default stays what the oracle tests saw."""
    lib = engine.lib
    x, eps, t, c = _inputs(64, 64, 3, flow="f32")
    dev = engine.device
    xd, ed, td = x.to(dev), eps.to(dev), t.to(dev)
    engine.set_prompts(c)
    out = {}
    try:
        for v in (1, 0, 2):
            assert lib.dm_set_option(b"tap_reuse", v) == 0
            out[v] = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
        assert lib.dm_set_option(b"tap_reuse", 1) == 0
        one = engine.score_conds(xd, ed[:1], td[:1], 2, latent_dtype=torch.float32).clone()      # rows (cond 0, draw 0), (cond 1, draw 0)
    finally:
        lib.dm_set_option(b"tap_reuse", 1)
    n = eps.shape[0]                                 # draws: row (cond k, draw i) = k * n + i
    assert torch.equal(one[0], out[1][0]) and torch.equal(one[1], out[1][n]), "a draw's loss depends on the batch with tap_reuse on"
    for v in (0, 2):
        rel = ((out[v] - out[1]).norm() / out[1].norm()).item()
        print(f"tap_reuse {v} vs 1: rel-L2 {rel:.2e}")
        assert rel < 2.5e-3, rel          # a pure reordering of fp32 partial sums: the oracle's own noise floor is 1.1-1.2e-3 (tests/test_oracle.py)
    assert not torch.equal(out[0], out[1])          # (the k order differs: equal bits would mean the option does nothing)


@pytest.mark.parametrize("h,w", [(64, 64), (16, 24)])
def test_up_fold_option_end_to_end(engine, sd15_weights_torch, h, w):
    """Option "up_fold" (Upsample2D + conv as four 2x2 convolutions on the source grid, igemm_pers_up.hip) end to end: the loss
    grids with the option on / off agree to the fp16 noise floor, a sample's bits do not depend on the batch it rides in with the
    option on, and — at the size the oracle reaches — the folded engine is no further from the fp32 oracle than the unfolded one."""
    lib = engine.lib
    x, eps, t, c = _inputs(h, w, 3, flow="f32")
    dev = engine.device
    xd, ed, td = x.to(dev), eps.to(dev), t.to(dev)
    engine.set_prompts(c)
    out = {}
    try:
        for v in (1, 0):
            assert lib.dm_set_option(b"up_fold", v) == 0
            out[v] = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
        assert lib.dm_set_option(b"up_fold", 1) == 0
        one = engine.score_conds(xd, ed[:1], td[:1], 2, latent_dtype=torch.float32).clone()
    finally:
        lib.dm_set_option(b"up_fold", 1)
    n = eps.shape[0]
    assert torch.equal(one[0], out[1][0]) and torch.equal(one[1], out[1][n]), "a draw's loss depends on the batch with up_fold on"
    rel = ((out[0] - out[1]).norm() / out[1].norm()).item()
    print(f"up_fold 0 vs 1 @{h}x{w}: loss rel-L2 {rel:.2e}")
    assert rel < 2.5e-3, rel                         # same bound as the pure re-orderings (the oracle's own noise floor is 1.1-1.2e-3)
    assert not torch.equal(out[0], out[1])           # (equal bits would mean the option does nothing)
    if h * w <= 16 * 24:
        nb, tb, cc, slots = _tile(eps, t, c)
        ref = R.compute_loss(sd15_weights_torch, x, nb, tb, cc, autocast=False, latent_dtype=torch.float32)
        e1 = ((out[1].cpu() - ref).norm() / ref.norm()).item()
        e0 = ((out[0].cpu() - ref).norm() / ref.norm()).item()
        print(f"vs the fp32 oracle: up_fold on {e1:.2e}, off {e0:.2e}")
        assert e1 <= 1.2 * e0, (e1, e0)


@pytest.mark.parametrize("h,w,n_draws", [(64, 64, 20), (16, 24, 3)])
def test_gn_epi_option_end_to_end(engine, h, w, n_draws):
    """Option "gn_epi" (r05, on: -0.3 ms per step): norm2's GroupNorm statistics as per-(64-row block,
    channel pair) sums written by conv1's epilogue where the persistent kernels run it (64 x 64: 20 draws x 2 prompts = 40 samples:
    the head of the 64-pixel-wide conv1 launches from the epilogue, the tail rows from the tensor; 16 x 24: HW % 64 == 0 at the
    first level only) — the loss grids agree with the statistics pass to a re-ordering of fp32 partial sums, and a draw's bits do
    not depend on the batch it rides in (one draw alone takes the 128-row tile: every block from the tensor)."""
    lib = engine.lib
    x, eps, t, c = _inputs(h, w, n_draws, flow="f32")
    dev = engine.device
    xd, ed, td = x.to(dev), eps.to(dev), t.to(dev)
    engine.set_prompts(c)
    out = {}
    try:
        for v in (1, 0):
            assert lib.dm_set_option(b"gn_epi", v) == 0
            out[v] = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
        assert lib.dm_set_option(b"gn_epi", 1) == 0
        one = engine.score_conds(xd, ed[:1], td[:1], 2, latent_dtype=torch.float32).clone()
        last = engine.score_conds(xd, ed[-2:], td[-2:], 2, latent_dtype=torch.float32).clone()
    finally:
        lib.dm_set_option(b"gn_epi", 1)
    n = eps.shape[0]
    assert torch.equal(one[0], out[1][0]) and torch.equal(one[1], out[1][n]), "a draw's loss depends on the batch with gn_epi on"
    assert torch.equal(last[1], out[1][n - 1]) and torch.equal(last[3], out[1][2 * n - 1])
    rel = ((out[0] - out[1]).norm() / out[1].norm()).item()
    print(f"gn_epi 0 vs 1 @{h}x{w}: loss rel-L2 {rel:.2e}")
    assert rel < 2.5e-3, rel
    assert not torch.equal(out[0], out[1])


@pytest.mark.parametrize("h,w,n_draws", [(64, 64, 4), (16, 24, 3)])
def test_gn_skip_option_end_to_end(engine, h, w, n_draws):
    """Option "gn_skip" (r05): the up path's norm1 over cat([x, skip]) merges the skip's GroupNorm sums kept from the down path instead of
    reading the skip again.  Loss grids on / off agree to a re-ordering of the statistics' sums; with the option on the shared-draw path
    (the skip's sums taken once per draw) equals the tiled batch bit for bit, and a draw's bits do not depend on the batch."""
    lib = engine.lib
    x, eps, t, c = _inputs(h, w, n_draws, flow="f32")
    dev = engine.device
    xd, ed, td = x.to(dev), eps.to(dev), t.to(dev)
    engine.set_prompts(c)
    nb, tb, cc, slots = _tile(eps, t, c)
    out = {}
    try:
        for v in (1, 0):
            assert lib.dm_set_option(b"gn_skip", v) == 0
            out[v] = engine.score_conds(xd, ed, td, 2, latent_dtype=torch.float32).clone()
        assert lib.dm_set_option(b"gn_skip", 1) == 0
        one = engine.score_conds(xd, ed[:1], td[:1], 2, latent_dtype=torch.float32).clone()
        tiled = engine.score(x, nb, tb, slots, latent_dtype=torch.float32).clone()
    finally:
        lib.dm_set_option(b"gn_skip", 1)
    n = eps.shape[0]
    assert torch.equal(one[0], out[1][0]) and torch.equal(one[1], out[1][n]), "a draw's loss depends on the batch with gn_skip on"
    assert torch.equal(tiled.to(out[1].device), out[1]), "shared-draw path and tiled batch differ with gn_skip on"
    rel = ((out[0] - out[1]).norm() / out[1].norm()).item()
    print(f"gn_skip 0 vs 1 @{h}x{w}: loss rel-L2 {rel:.2e}")
    assert rel < 2.5e-3, rel
    assert not torch.equal(out[0], out[1])
