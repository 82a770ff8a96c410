#!/usr/bin/env python
"""T(x|c) of the fp16 engine against the EXACT-fp32 evaluation of the same U-Net at the BASELINE configuration itself
(configs[1]: 64 x 64 latent, N = 10 draws x 2 prompts per image), both on the GPU: the fp32 side is the fp32 net (dm_f32_*,
checked against the CPU oracle's autocast=False arithmetic by tests/test_gpu_f32.py): dm_f32_score = compute.py:95-102
with no autocast.  tools/t_deviation.py is the same table against the CPU oracle, which reaches 32 x 32 only.

    python tools/t_deviation_gpu.py [n_images] > profiles/r04_T_deviation_baseline_size_fp32.txt
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
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    N, hw = 10, 64
    sdn = synth.synth_state_dict(seed=0, dtype=np.float16)
    e16, e32 = UNetEngine(0), UNetEngineF32(0)
    e16.load_state_dict(sdn)
    e32.load_state_dict(sdn)
    dev = e16.device
    sc = TypicalityScorer(e16, seed=42, N=N, t_min=0.1, t_max=0.7)
    xs, _, _, c = synth.synth_inputs(n_img, 1, hw, hw, latent_dtype=np.float32)
    xs, c = torch.from_numpy(xs), torch.from_numpy(c)
    e32.set_prompts(c.float())
    rows = []
    for i in range(n_img):
        x = xs[i:i + 1]
        noises, ts = sc.draw(x.shape)
        grid = sc.compute_losses(x, c, noises=noises, timesteps=ts, to_host=False).float()       # [N,2,4,h,w] (fp16 values)
        ref = e32.score_conds(x, noises, ts, 2)                                          # dm_f32_score, cond-major rows
        ref = ref.view(2, N, 4, hw, hw).transpose(0, 1)
        T = (grid[:, 1] - grid[:, 0]).mean().item()
        T32 = (ref[:, 1] - ref[:, 0]).double().mean().item()
        ml = ref.mean().item()
        rl = ((grid - ref).double().norm() / ref.double().norm()).item()
        rows.append((rl, abs(T - T32) / abs(T32), abs(T - T32) / ml))
        print(f"image {i:2d}: T fp16 engine {T:+.6e}  exact fp32 {T32:+.6e}  mean loss {ml:.4f}   grid rel-L2 {rl:.2e}   "
              f"|dT|/|T| {rows[-1][1]:.2e}   |dT|/mean loss {rows[-1][2]:.2e}")
    r = np.array(rows)
    print(f"{n_img} images @64x64, N = 10 x 2 prompts: grid rel-L2 max {r[:, 0].max():.2e} mean {r[:, 0].mean():.2e};  |dT|/|T| max {r[:, 1].max():.2e} "
          f"median {np.median(r[:, 1]):.2e};  |dT|/mean loss max {r[:, 2].max():.2e} median {np.median(r[:, 2]):.2e}")


if __name__ == "__main__":
    main()
