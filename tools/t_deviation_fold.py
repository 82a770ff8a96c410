#!/usr/bin/env python
"""Does an arithmetic switch move T(x|c)?  Paired comparison at the BASELINE configuration (64 x 64 latent, N = 10 draws x 2 prompts):
the fp16 engine with the switch ON and OFF on the same images and draws, both against ONE exact-fp32 evaluation (the fp32 net with the
switch off).  A rewrite that perturbs the weights (up_fold: summed taps rounded once) acts like a slightly different model — its
effect on T does not average out over the draws the way rounding noise does — so the signed, paired statistics matter, not only the
medians of tools/t_deviation_gpu.py.

    python tools/t_deviation_fold.py [n_images] [option[,option...]=up_fold] > profiles/r04_T_deviation_up_fold_paired.txt

An option may carry its two values (`attn_pipe=1:3`: "on" = 1, "off" = 3; default 1:0) — r06: the scores of head_dim-40 self-attention on
32x32x16 MFMAs (attn_pipe 1) against r05's kernel (attn_pipe 3).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from diff_mining_amd import synth  # noqa: E402
from diff_mining_amd.engine import UNetEngine, UNetEngineF32  # noqa: E402
from diff_mining_amd.typicality import TypicalityScorer  # noqa: E402


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    spec = (sys.argv[2] if len(sys.argv) > 2 else "up_fold").split(",")                          # several switches: toggled together
    opts = [o.split("=")[0].encode() for o in spec]
    vals = {o.split("=")[0].encode(): (tuple(int(v) for v in o.split("=")[1].split(":")) if "=" in o else (1, 0)) for o in spec}
    opt = "+".join(spec).encode()
    N, hw = 10, 64
    sdn = synth.synth_state_dict(seed=0, dtype=np.float16)
    e16, e32 = UNetEngine(0), UNetEngineF32(0)
    e16.load_state_dict(sdn)
    e32.load_state_dict(sdn)
    lib = e16.lib
    sc = TypicalityScorer(e16, seed=42, N=N, t_min=0.1, t_max=0.7)
    xs, _, _, c = synth.synth_inputs(n_img, 1, hw, hw, latent_dtype=np.float32)
    xs, c = torch.from_numpy(xs), torch.from_numpy(c)
    e32.set_prompts(c.float())
    rows = []
    for i in range(n_img):
        x = xs[i:i + 1]
        noises, ts = sc.draw(x.shape)
        g = {}
        for v in (1, 0):
            for o in opts:
                assert lib.dm_set_option(o, vals[o][0] if v else vals[o][1]) == 0
            g[v] = sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False).float()
        ref = e32.score_conds(x, noises, ts, 2).view(2, N, 4, hw, hw).transpose(0, 1)        # fp32 net, switch off
        for o in opts:
            lib.dm_set_option(o, vals[o][0])
        T32 = (ref[:, 1] - ref[:, 0]).double().mean().item()
        ml = ref.mean().item()
        T = {v: (g[v][:, 1] - g[v][:, 0]).double().mean().item() for v in (1, 0)}
        rl = {v: ((g[v] - ref).double().norm() / ref.double().norm()).item() for v in (1, 0)}
        rows.append((T32, ml, T[1] - T32, T[0] - T32, rl[1], rl[0]))
        print(f"image {i:2d}: T fp32 {T32:+.6e}  dT on {T[1] - T32:+.3e} off {T[0] - T32:+.3e}  (of the mean loss {ml:.4f}: {(T[1] - T32) / ml:+.2e} / {(T[0] - T32) / ml:+.2e})"
              f"   grid rel-L2 on {rl[1]:.3e} off {rl[0]:.3e}")
    r = np.array(rows)
    T32, ml, d1, d0 = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    for name, d in (("on ", d1), ("off", d0)):
        rel, rml = d / np.abs(T32), d / ml
        print(f"{opt.decode()} {name}: |dT|/|T| rms {np.sqrt((rel ** 2).mean()):.2e} median {np.median(np.abs(rel)):.2e} max {np.abs(rel).max():.2e};  "
              f"dT/mean loss rms {np.sqrt((rml ** 2).mean()):.2e}  signed mean {rml.mean():+.2e} +- {rml.std(ddof=1) / np.sqrt(len(d)):.1e} (standard error);  "
              f"grid rel-L2 mean {r[:, 4 if name == 'on ' else 5].mean():.3e}")
    dd = (d1 - d0) / ml
    print(f"paired (on - off) / mean loss: mean {dd.mean():+.2e} +- {dd.std(ddof=1) / np.sqrt(len(dd)):.1e}, rms {np.sqrt((dd ** 2).mean()):.2e}   "
          f"[{n_img} images @64x64, N = 10 x 2 prompts]")


if __name__ == "__main__":
    main()
