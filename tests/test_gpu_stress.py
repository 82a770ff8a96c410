"""The engine's algebraic rewrites OFF the benign operating point.

r03 added four rewrites that move rounding points — LayerNorm folded into the following GEMM (`ln_fold`: rstd (acc - mean sum W')),
GroupNorm folded into per-sample weights (`gn_fold`: fp16(W diag(a_n)), 64x64 level), `ff.net.2` + `proj_out` pre-multiplied
(`ff_fold`), `conv_shortcut` folded into `conv2` (`sc_fold`); r04: `Upsample2D` + conv as four 2x2 convolutions with pre-summed
taps (`up_fold`) — each validated only on weights that give zero-mean O(1)
activations.  Here the same U-Net runs on `tests/stress_weights.py`: output-channel scales over two decades, x50 outlier channels,
|mean| / std of 5-20 at the LayerNorm inputs and the ResNets' inner GroupNorm, 9 (median) / 18 (max) at the first transformer's
GroupNorm — the place where a folded form cancels a large common mode against a rounded operand.

For every size: eps_hat of the engine with every fold ON (the default) and with each fold OFF, against the fp32 oracle (ground
truth) and the fp16-autocast oracle.  Asserted: folded error <= unfused error x 1.2 (vs fp32), and the absolute level.
Reference call being stood in for: `unet(noisy_latents, t, c).sample`, diffmining/typicality/compute.py:100 (real checkpoints:
README.md:52-56).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diff_mining_amd import synth  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from tests import stress_weights as S  # noqa: E402

FOLDS = ("ln_fold", "gn_fold", "ff_fold", "sc_fold", "up_fold")


@pytest.fixture(scope="module")
def stress_sd():
    return S.build()


@pytest.fixture(scope="module")
def stress_engine(stress_sd):
    from diff_mining_amd.engine import UNetEngine
    e = UNetEngine(0)
    e.load_state_dict({k: v.numpy().astype(np.float16) for k, v in stress_sd.items()})
    yield e
    e.close()


def _case(hw):
    x, eps, t, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, 1, hw, hw, latent_dtype=np.float32))
    nb, tb = torch.cat([eps] * 2), torch.cat([t] * 2)
    noisy = R.add_noise(x.expand(2, -1, -1, -1), nb, tb)
    return noisy, tb, c


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def test_stress_operating_point_is_off_benign(stress_sd, sd15_weights_torch):
    """What the stress weights do to the normalisation inputs (fp32 oracle, 16x16), next to the benign weights."""
    st, bn = S.operating_point(stress_sd, 16), S.operating_point(sd15_weights_torch, 16)
    ln = np.array([v[0] for k, v in st.items() if "transformer_blocks" in k])
    n2 = np.array([v[0] for k, v in st.items() if k.endswith("norm2") and "resnets" in k])
    bln = np.array([v[0] for k, v in bn.items() if "transformer_blocks" in k])
    mx = max(v[2] for v in st.values())
    print(f"stress: LayerNorm inputs |mean|/std median {np.median(ln):.1f} (q10 {np.quantile(ln, .1):.1f}, q90 {np.quantile(ln, .9):.1f}); "
          f"resnet norm2 {np.median(n2):.1f}; first transformer GroupNorm {st['down_blocks.0.attentions.0.norm'][0]:.1f} (max group "
          f"{st['down_blocks.0.attentions.0.norm'][1]:.1f}); max |activation| {mx:.0f}   |   benign: LayerNorm inputs {np.median(bln):.2f}")
    assert np.quantile(ln, .1) > 4 and np.median(ln) > 8 and np.median(n2) > 5
    assert st["down_blocks.0.attentions.0.norm"][0] > 5
    assert mx < 65504 / 8                        # fp16 headroom for the larger latents
    assert np.median(bln) < 1.0                  # the benign weights really are benign


@pytest.mark.parametrize("hw", [8, 16, 32, 64])
def test_folds_off_the_benign_operating_point(stress_engine, stress_sd, hw):
    eng = stress_engine
    lib = eng.lib
    noisy, tb, c = _case(hw)
    cc = torch.cat([c[0:1], c[1:2]]).float()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        p32 = R.unet_forward(stress_sd, noisy, tb, cc, autocast=False).float()
        pac = R.unet_forward(stress_sd, noisy, tb, cc, autocast=True).float()
    assert torch.isfinite(pac).all()
    eng.set_prompts(c)
    slots = torch.tensor([0, 1], dtype=torch.int32)
    x16 = noisy.half()

    def run(off=()):
        try:
            for f in FOLDS:
                assert lib.dm_set_option(f.encode(), 0 if f in off else 1) == 0
            return eng.unet(x16, tb, slots).float().cpu()
        finally:
            for f in FOLDS:
                lib.dm_set_option(f.encode(), 1)
    res = {"all folds on": run(), "all folds off": run(FOLDS)}
    for f in FOLDS:
        res[f"{f} off"] = run((f,))
    oa = _rel(pac, p32)
    print(f"\n[stress {hw}x{hw}] autocast oracle vs fp32 oracle: {oa:.2e}")
    err = {}
    for name, p in res.items():
        assert torch.isfinite(p).all(), name
        err[name] = (_rel(p, p32), _rel(p, pac))
        print(f"[stress {hw}x{hw}] engine, {name:14s}: eps_hat rel-L2 vs fp32 oracle {err[name][0]:.2e}  vs autocast oracle {err[name][1]:.2e}")
    on, off = err["all folds on"][0], err["all folds off"][0]
    # the bar: a folded form may not lose more than 20 % against the unfused kernels anywhere off the benign point
    assert on <= 1.2 * off, (on, off)
    for f in FOLDS:
        assert on <= 1.2 * err[f"{f} off"][0], (f, on, err[f"{f} off"][0])
    # and the absolute level: the engine is as close to exact arithmetic as fp16 autocast itself is (x 1.5)
    assert on <= 1.5 * oa, (on, oa)
