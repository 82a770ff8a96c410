#!/usr/bin/env python
"""Two CPU-only measurements of the fp16-autocast oracle against itself.  Test infrastructure: runs the oracle.

  floor     T(x|c) of the autocast oracle under pure re-orderings of fp32 partial sums, at the BASELINE draw count: N = 10
            draws x 2 prompts, 32x32 latents, two images, the seed-42 draws of `D.compute_losses` (compute.py:139-141).
            Variants: all threads (base) / one thread / channels-last convolutions / hidden channels permuted
            (tests/test_oracle.py::_reparametrised: an exact identity of the function).  Prints |dT|/|T| and |dT|/mean loss
            per image and variant, writes tests/golden/oracle_T_floor.json — the bound
            tests/test_gpu_e2e.py::test_score_at_baseline_draw_count holds the engine to (engine <= 1.5 x the oracle's own
            spread) — and profiles/r04_oracle_T_noise_floor.txt.
  ablation  which fp16 roundings of the emulation produce the autocast-vs-fp32 distance: every rounding site of
            oracle/unet_ref.py switched off alone (and all but one off), eps_hat / loss rel-L2 vs the fp32 oracle at 8x8, 16x16
            (and 32x32 with --big) -> profiles/r04_oracle_rounding_ablation.txt.

    python tools/oracle_noise.py floor|ablation [--big]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from diff_mining_amd import synth  # noqa: E402
from oracle import unet_ref as R  # noqa: E402
from tests.test_oracle import _reparametrised  # noqa: E402

SITES = ["in", "out", "res", "temb", "p", "o", "geglu", "emb", "x"]


def weights():
    return {k: torch.from_numpy(v).float() for k, v in synth.synth_state_dict(seed=0, dtype=np.float16).items()}


def floor_inputs(hw=32, n_img=2, N=10):
    """Exactly what tests/test_gpu_e2e.py::test_score_at_baseline_draw_count scores."""
    x, _, _, c = synth.synth_inputs(n_img, 1, hw, hw, latent_dtype=np.float32)
    noises, ts = R.draw_noise_and_timesteps((1, 4, hw, hw), N, 0.1, 0.7, seed=42)     # the test's TypicalityScorer(seed=42, t_min=0.1, t_max=0.7)
    return torch.from_numpy(x), noises, ts, torch.from_numpy(c).float()


def grid_T(sd, x, noises, ts, c, threads=None, cl=False):
    old = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        if cl:
            sd = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()}
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            g = R.compute_losses(sd, x, c, noises, ts, B=5, autocast=True, latent_dtype=torch.float32)
        return g, R.typicality_scalar(g).item(), g.float().mean().item()
    finally:
        torch.set_num_threads(old)


def floor(hw=32, n_img=2, N=10):
    sd = weights()
    x, noises, ts, c = floor_inputs(hw, n_img, N)
    sdp = _reparametrised(sd)
    lines, rec = [], {"latent": hw, "n_draws": N, "n_cond": 2, "images": []}
    for i in range(n_img):
        t0 = time.time()
        base_g, base_T, mean_loss = grid_T(sd, x[i:i + 1], noises, ts, c)
        var = {"1 thread": grid_T(sd, x[i:i + 1], noises, ts, c, threads=1),
               "channels-last": grid_T(sd, x[i:i + 1], noises, ts, c, cl=True),
               "hidden channels permuted": grid_T(sdp, x[i:i + 1], noises, ts, c)}
        img = {"T": base_T, "mean_loss": mean_loss, "variants": {}}
        for name, (g, T, _) in var.items():
            rl = ((g.float() - base_g.float()).norm() / base_g.float().norm()).item()
            dT = abs(T - base_T)
            img["variants"][name] = {"T": T, "dT_over_T": dT / abs(base_T), "dT_over_mean_loss": dT / mean_loss, "grid_rel_l2": rl}
            lines.append(f"image {i} ({hw}x{hw}, N = {N} x 2 prompts): autocast oracle vs itself, {name:26s}: T {T:+.6e} vs {base_T:+.6e}  "
                         f"|dT|/|T| {dT / abs(base_T):.2e}  |dT|/mean loss {dT / mean_loss:.2e}  grid rel-L2 {rl:.2e}")
        rec["images"].append(img)
        print(f"[image {i}: {time.time() - t0:.0f} s]", file=sys.stderr)
    rec["max_dT_over_T"] = max(v["dT_over_T"] for im in rec["images"] for v in im["variants"].values())
    rec["max_dT_over_mean_loss"] = max(v["dT_over_mean_loss"] for im in rec["images"] for v in im["variants"].values())
    lines.append(f"max over images and re-orderings: |dT|/|T| {rec['max_dT_over_T']:.2e}  |dT|/mean loss {rec['max_dT_over_mean_loss']:.2e}")
    print("\n".join(lines))
    return rec, lines


def ablation(sizes=(8, 16)):
    sd = weights()
    lines = []
    for hw in sizes:
        x, eps, t, c = (torch.from_numpy(a) for a in synth.synth_inputs(1, 1, hw, hw, latent_dtype=np.float32))
        nb, tb = torch.cat([eps] * 2), torch.cat([t] * 2)
        cc = torch.cat([c[0:1], c[1:2]]).float()
        noisy = R.add_noise(x.expand(2, -1, -1, -1), nb, tb)

        def run(ac):
            with torch.no_grad():
                p = R.unet_forward(sd, noisy, tb, cc, autocast=ac).float()
            return p, torch.nn.functional.mse_loss(p, nb, reduction="none")
        p32, l32 = run(False)

        def dist(off):
            R.ROUND_OFF.clear()
            R.ROUND_OFF.update(off)
            try:
                p, l = run(True)
            finally:
                R.ROUND_OFF.clear()
            return ((p - p32).norm() / p32.norm()).item(), ((l - l32).norm() / l32.norm()).item()
        full = dist(())
        lines.append(f"latent {hw}x{hw}: autocast oracle vs fp32 oracle, all roundings on: eps_hat rel-L2 {full[0]:.2e}  loss rel-L2 {full[1]:.2e}")
        for s in SITES:
            a = dist((s,))
            b = dist([o for o in SITES if o != s])
            lines.append(f"latent {hw}x{hw}:   site {s:6s} OFF alone: eps_hat {a[0]:.2e} ({a[0] / full[0]:4.0%} of all-on)  loss {a[1]:.2e}   |   "
                         f"ONLY {s:6s} on: eps_hat {b[0]:.2e} ({b[0] / full[0]:4.0%})  loss {b[1]:.2e}")
        a = dist(("res", "temb"))
        lines.append(f"latent {hw}x{hw}:   fp32 residual stream (res + temb off): eps_hat {a[0]:.2e} ({a[0] / full[0]:4.0%})  loss {a[1]:.2e}")
        a = dist(("res", "temb", "out"))
        lines.append(f"latent {hw}x{hw}:   res + temb + out off (only operands rounded): eps_hat {a[0]:.2e} ({a[0] / full[0]:4.0%})  loss {a[1]:.2e}")
    print("\n".join(lines))
    return lines


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "floor"
    if what == "floor":
        rec, lines = floor()
        with open(os.path.join(ROOT, "tests", "golden", "oracle_T_floor.json"), "w") as f:
            json.dump(rec, f, indent=1)
        with open(os.path.join(ROOT, "profiles", "r04_oracle_T_noise_floor.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    else:
        lines = ablation((8, 16, 32) if "--big" in sys.argv else (8, 16))
        with open(os.path.join(ROOT, "profiles", "r04_oracle_rounding_ablation.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
